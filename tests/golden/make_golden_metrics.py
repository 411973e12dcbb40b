"""Golden values of the evaluation metrics from the REFERENCE's own src/metrics/metrics.py (authoring container only).

    python tests/golden/make_golden_metrics.py   ->  tests/golden/metrics.npz

metrics.py imports deeptime (TICA) at module level; it is stubbed with an empty module because js_tica is not exercised
(it needs the real estimator).  Inputs: two seeded float32 CA ensembles per case -- a jittered helix 'target' and a noisier
'pred' -- as extract_backbone_coords would return them; outputs: validity, bonding_validity, js_pwd, js_rg (the dict values the
reference returns, rounded to 4 decimals by its own code) plus un-rounded per-channel material for a bit-level check of the
histogram path."""
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import _ref_import  # noqa: E402

_ref_import.install()
dt = types.ModuleType("deeptime"); dd = types.ModuleType("deeptime.decomposition"); dd.TICA = object
sys.modules.setdefault("deeptime", dt); sys.modules.setdefault("deeptime.decomposition", dd)
from scipy.spatial import distance  # noqa: E402
from src.metrics import metrics as M  # noqa: E402


def ensemble(rng, R, L, jitter, clash_frac=0.0, break_frac=0.0):
    k = np.arange(L)
    base = np.stack([2.3 * np.cos(1.745 * k), 2.3 * np.sin(1.745 * k), 1.5 * k], -1)
    x = base[None] + jitter * rng.normal(size=(R, L, 3))
    scale = 1.0 + 0.15 * rng.normal(size=(R, 1, 1))          # spread the radius of gyration
    x = x * scale
    for s in range(R):
        if rng.random() < clash_frac:
            i, j = rng.choice(L, 2, replace=False)
            x[s, j] = x[s, i] + 0.5                            # a steric clash
        if rng.random() < break_frac:
            x[s, L // 2:] += 6.0                               # a broken bond
    return x.astype(np.float32)


def main():
    rng = np.random.default_rng(7)
    out = {}
    for tag, (Rt, R, L) in {"a": (200, 150, 12), "b": (64, 300, 40), "c": (1, 50, 9)}.items():
        tgt, pred = ensemble(rng, Rt, L, 0.3), ensemble(rng, R, L, 0.8, clash_frac=0.3, break_frac=0.2)
        d = {"target": tgt, "pred": pred}
        out[f"{tag}_target"], out[f"{tag}_pred"] = tgt, pred
        out[f"{tag}_validity"] = np.array([M.validity(d)[k] for k in ("target", "pred")])
        out[f"{tag}_bonding"] = np.array([M.bonding_validity(d)[k] for k in ("target", "pred")])
        out[f"{tag}_js_pwd"] = np.array(M.js_pwd(d)["pred"])
        out[f"{tag}_js_rg"] = np.array(M.js_rg(d)["pred"])
        # un-rounded per-channel Jensen-Shannon distances of js_pwd (metrics.py:152-162 re-typed around the reference's helpers)
        pwd = {k: M.pairwise_distance_ca(v, k=3) for k, v in d.items()}
        lo, hi = pwd["target"].min(axis=0), pwd["target"].max(axis=0)
        binned = {k: np.apply_along_axis(lambda a: np.histogram(a[:-2], bins=50, range=(a[-2], a[-1]))[0] + M.PSEUDO_C, 0,
                                         np.concatenate([v, lo[None], hi[None]], axis=0)) for k, v in pwd.items()}
        out[f"{tag}_js_pwd_channels"] = distance.jensenshannon(binned["pred"], binned["target"], axis=0)
        out[f"{tag}_rg_pred"] = M.radius_of_gyration(pred)
    path = os.path.join(HERE, "metrics.npz")
    np.savez_compressed(path, **out)
    print(f"metrics.npz: {os.path.getsize(path)/1024:.1f} KiB", {k: out[k] for k in out if k.endswith(("validity", "bonding", "js_pwd", "js_rg"))})


if __name__ == "__main__":
    main()
