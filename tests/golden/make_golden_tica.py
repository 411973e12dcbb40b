"""Golden values for the `js_tica` column and for per-sample `weights=` in js_pwd / js_rg / js_tica (authoring container only).

    python tests/golden/make_golden_tica.py   ->  tests/golden/tica.npz

Everything except the estimator is the REFERENCE's own code: src/metrics/metrics.py is imported from /root/reference and its
js_tica / js_pwd / js_rg are called.  The estimator itself, deeptime.decomposition.TICA (deeptime==0.4.4, environment.yml:184), is a
third-party dependency that is absent here and from /root/reference: `oracle/tica.py` -- a restatement of its published algorithm
with that release's conventions, NOT a run of deeptime -- is installed under its module name, so the reference's js_tica drives it
exactly as it would drive the real one (fit on the reference ensemble's pairwise distances, transform every ensemble, histograms
over the reference's range, Jensen-Shannon).  js_tica is therefore pinned to "reference code around the restated estimator";
js_pwd / js_rg with weights are the reference alone.
Inputs: a `target` trajectory of a jittered helix whose last residues swing slowly between two states (so that TICA has a slow
mode to find), a `pred` ensemble with other populations, float64 sample weights for `pred`."""
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)
import _ref_import  # noqa: E402

_ref_import.install()
from oracle import tica as OT  # noqa: E402

dt = types.ModuleType("deeptime"); dd = types.ModuleType("deeptime.decomposition"); dd.TICA = OT.TICA
sys.modules["deeptime"], sys.modules["deeptime.decomposition"] = dt, dd
from src.metrics import metrics as M  # noqa: E402


def trajectory(rng, T, L, p_stay, jitter):
    """Two-state hinge: frames of a helix whose second half is rotated by +-0.5 rad about x, switching with probability 1 - p_stay."""
    k = np.arange(L)
    base = np.stack([2.3 * np.cos(1.745 * k), 2.3 * np.sin(1.745 * k), 1.5 * k], -1)
    state, frames = 1.0, []
    slow = 0.0
    for _ in range(T):
        if rng.random() > p_stay:
            state = -state
        slow = 0.9 * slow + 0.1 * state                      # relaxes towards the state: a slow coordinate
        a = 0.5 * slow
        rot = np.array([[1, 0, 0], [0, np.cos(a), -np.sin(a)], [0, np.sin(a), np.cos(a)]])
        x = base.copy()
        x[L // 2:] = (x[L // 2:] - x[L // 2]) @ rot.T + x[L // 2]
        frames.append(x + jitter * rng.normal(size=(L, 3)))
    return np.asarray(frames, dtype=np.float32)


def main():
    rng = np.random.default_rng(11)
    out = {}
    for tag, (T, R, L, lag) in {"a": (400, 300, 12, 20), "b": (260, 90, 10, 5)}.items():
        tgt, pred = trajectory(rng, T, L, 0.97, 0.15), trajectory(rng, R, L, 0.90, 0.35)
        w = rng.gamma(2.0, 1.0, size=R)
        d = {"target": tgt, "pred": pred}
        out[f"{tag}_target"], out[f"{tag}_pred"], out[f"{tag}_weights"], out[f"{tag}_lag"] = tgt, pred, w, np.array(lag)
        res, tics = M.js_tica(d, ref_key="target", lagtime=lag)
        out[f"{tag}_js_tica"], out[f"{tag}_tic_target"], out[f"{tag}_tic_pred"] = np.array(res["pred"]), tics["target"], tics["pred"]
        out[f"{tag}_js_tica_w"] = np.array(M.js_tica(d, ref_key="target", lagtime=lag, weights={"pred": w.copy()})[0]["pred"])
        out[f"{tag}_js_pwd_w"] = np.array(M.js_pwd(d, ref_key="target", weights={"pred": w.copy()})["pred"])
        out[f"{tag}_js_rg_w"] = np.array(M.js_rg(d, ref_key="target", weights={"pred": w.copy()})["pred"])
        # un-rounded material: the estimator's eigenvalues, per-channel weighted Jensen-Shannon distances of js_pwd
        est = OT.TICA(dim=2, lagtime=lag).fit(M.pairwise_distance_ca(tgt))
        out[f"{tag}_tica_eigenvalues"] = est.eigenvalues[:4]
        from scipy.spatial import distance
        pwd = {k: M.pairwise_distance_ca(v, k=3) for k, v in d.items()}
        lo, hi = pwd["target"].min(axis=0), pwd["target"].max(axis=0)
        ww = {"target": np.ones(T), "pred": w}
        binned = {k: np.apply_along_axis(lambda a, k=k: np.histogram(a[:-2], bins=50, weights=ww[k], range=(a[-2], a[-1]))[0] + M.PSEUDO_C, 0,
                                         np.concatenate([v, lo[None], hi[None]], axis=0)) for k, v in pwd.items()}
        out[f"{tag}_js_pwd_w_channels"] = distance.jensenshannon(binned["pred"], binned["target"], axis=0)
    path = os.path.join(HERE, "tica.npz")
    np.savez_compressed(path, **out)
    print(f"tica.npz: {os.path.getsize(path)/1024:.1f} KiB", {k: out[k] for k in out if "js_" in k and "channels" not in k},
          {k: out[k] for k in out if "eigen" in k})


if __name__ == "__main__":
    main()
