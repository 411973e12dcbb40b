"""Ensemble metrics of the evaluation step on the device (reference: src/metrics/metrics.py; driver src/eval.py:47-99).

Same function names, arguments and return values (dicts keyed like the input, rounded to 4 decimals) as the reference for
``validity`` (:108-121), ``bonding_validity`` (:124-137), ``js_pwd`` (:140-166) and ``js_rg`` (:203-224).  The N^2 x R work --
pairwise CA distances, per-channel histograms and Jensen-Shannon distances, clash counts, radii of gyration -- runs in two HIP
kernels (csrc/ensemble_metrics.hip) straight on the coordinates the sampler just produced (or on arrays read back from PDB
files); numpy only finishes the O(R) / O(bins) tails.  ``js_tica`` (:169-200): pairwise distances on the device, then the TICA
projection -- deeptime's estimator when it is installed (the reference's), otherwise the same estimator in numpy (``tica_fit``:
deeptime 0.4.4's conventions -- reversible covariances, absolute 1e-6 cut-off, eigenpairs by descending magnitude, canonical signs,
kinetic-map scaling -- so that the projections themselves, not only the score, are the reference's).  Per-sample ``weights=``
(:139-150,178-179,206-208) are histogram weights on both paths.
"""
from __future__ import annotations

import numpy as np
import torch

from .. import ops

EPS = 1e-12
PSEUDO_C = 1e-6


def _dev(x) -> torch.Tensor:
    t = x if torch.is_tensor(x) else torch.as_tensor(np.asarray(x))
    if t.ndim == 2:
        t = t[None]
    assert t.ndim == 3 and t.shape[-1] == 3, f"CA coords should be 2D or 3D, got {tuple(t.shape)}"
    return t.to("cuda", torch.float32).contiguous()


def _js(p: np.ndarray, q: np.ndarray) -> float:
    """scipy.spatial.distance.jensenshannon on two 1-D vectors (natural log)."""
    p = p / p.sum(); q = q / q.sum()
    m = (p + q) / 2.0
    rel = lambda x, y: np.where(x > 0, x * np.log(np.where(x > 0, x, 1.0) / y), 0.0)  # noqa: E731
    return float(np.sqrt((rel(p, m).sum() + rel(q, m).sum()) / 2.0))


def validity(ca_coords_dict, ca_vdw_radius=1.7, allowable_overlap=0.4, k_exclusion=0):
    bar = 2 * ca_vdw_radius - allowable_overlap
    out = {}
    for k, v in ca_coords_dict.items():
        n_clash, _, _ = ops.ca_sample_stats(_dev(v), bar, k_exclusion)
        out[k] = np.around(1.0 - float((n_clash > 0).double().mean()), decimals=4)
    return out


def bonding_validity(ca_coords_dict, ref_key="target", eps=1e-6):
    adj = {k: ops.ca_sample_stats(_dev(v))[1] for k, v in ca_coords_dict.items()}
    thres = adj[ref_key].max() + 1e-6          # float32 arithmetic, as the reference forms it (metrics.py:133)
    return {k: np.around(float((a < thres).sum()) / len(a), decimals=4) for k, a in adj.items()}


def _weights(weights, ca_coords_dict):
    """The reference's default (metrics.py:148-150): ones for every ensemble without an entry.  -> {k: float64 [len(v)]}"""
    w = {k: np.asarray(v, dtype=np.float64) for k, v in (weights or {}).items()}
    for k, v in ca_coords_dict.items():
        w.setdefault(k, np.ones(len(v)))
        if w[k].shape != (len(v),):
            raise ValueError(f"weights[{k!r}] has shape {w[k].shape} for {len(v)} samples")
    return w


def js_pwd(ca_coords_dict, ref_key="target", n_bins=50, pwd_offset=3, weights=None):
    ref = _dev(ca_coords_dict[ref_key])
    wd = None
    if weights:   # float64 per-sample weights into the device histograms
        wd = {k: torch.as_tensor(v, device="cuda") for k, v in _weights(weights, ca_coords_dict).items()}
    out = {k: np.around(float(ops.ca_pwd_js(ref, _dev(v), pwd_offset, n_bins, PSEUDO_C, ref_weights=wd[ref_key] if wd else None,
                                            pred_weights=wd[k] if wd else None).mean()), decimals=4)
           for k, v in ca_coords_dict.items() if k != ref_key}
    out[ref_key] = 0.0
    return out


def radius_of_gyration(coords):
    return ops.ca_sample_stats(_dev(coords))[2].cpu().numpy()


def js_rg(ca_coords_dict, ref_key="target", n_bins=50, weights=None):
    w = _weights(weights, ca_coords_dict)
    # the reference's Rg is float64 (float32 squared distances x float64 weights, metrics.py:62-78) and so are its histogram edges
    rg = {k: np.asarray(radius_of_gyration(v), dtype=np.float64) for k, v in ca_coords_dict.items()}
    d_min, d_max = rg[ref_key].min(), rg[ref_key].max()
    binned = {k: np.histogram(v, bins=n_bins, weights=w[k], range=(d_min, d_max))[0] + PSEUDO_C for k, v in rg.items()}
    out = {k: np.around(_js(v, binned[ref_key]), decimals=4) for k, v in binned.items() if k != ref_key}
    out[ref_key] = 0.0
    return out


def pairwise_distance_ca(coords, k=1) -> np.ndarray:
    """reference :38-50: upper-triangular CA distances [B, (L - k)(L - k + 1)/2], float32, computed on the device in numpy's own
    float32 arithmetic (s2s_ca_pairwise_distances): bit for bit the reference's features."""
    return ops.ca_pairwise_distances(_dev(coords), k).cpu().numpy()


def tica_fit(x: np.ndarray, lagtime: int, dim: int = 2, epsilon: float = 1e-6):
    """TICA (time-lagged independent component analysis) of a trajectory x [T, D] as deeptime 0.4.4's ``TICA(dim, lagtime)`` -- the
    estimator the reference calls (metrics.py:175; environment.yml:184) -- computes it: reversible estimate (mean and covariances
    symmetrised over the (x_t, x_{t+lag}) pairs, no Bessel correction), C00 whitened on the eigenvectors whose eigenvalue magnitude
    reaches the ABSOLUTE cut-off ``epsilon`` (raised above the magnitude of the most negative eigenvalue when rounding produced one -- a
    relative cut-off would keep a different subspace for rank-deficient pairwise-distance features, whose largest eigenvalue is
    1e2 .. 1e4 A^2), symmetric eigenproblem of the whitened time-lagged covariance, eigenpairs by DESCENDING MAGNITUDE (a negative
    eigenvalue can rank second), canonical signs (every vector's largest-magnitude entry positive), kinetic-map scaling (vector x
    eigenvalue).  -> (mean [D], projection [D, dim]); transform = (x - mean) @ proj.
    Pinned by an analytic two-state process and against the restatement in oracle/tica.py (tests/test_host_cpu.py)."""
    import scipy.linalg

    x = np.asarray(x, dtype=np.float64)
    if x.shape[0] <= lagtime:
        raise ValueError(f"js_tica: {x.shape[0]} frames are not enough for lagtime {lagtime}")

    def by_magnitude(vals, vecs):
        order = np.argsort(np.abs(vals))[::-1]
        return vals[order], vecs[:, order]

    def canonical(vecs):
        top = np.argmax(np.abs(vecs), axis=0)
        return vecs * np.sign(vecs[top, np.arange(vecs.shape[1])])[None, :]

    x0, xt = x[:-lagtime], x[lagtime:]
    mean = 0.5 * (x0.mean(0) + xt.mean(0))
    a, b = x0 - mean, xt - mean
    n = 2.0 * a.shape[0]
    c00 = (a.T @ a + b.T @ b) / n
    c0t = (a.T @ b + b.T @ a) / n
    w, v = by_magnitude(*scipy.linalg.eigh(c00))
    cut = max(epsilon, -w.min() + 1e-16) if w.min() < 0 else epsilon
    m = len(w) - int(np.searchsorted(np.abs(w)[::-1], cut))
    if m == 0:
        raise ValueError("js_tica: the reference ensemble has no variance above the cut-off")
    white = canonical(v[:, :m]) / np.sqrt(w[:m])[None, :]     # [D, m]: white.T C00 white = I
    lam, u = by_magnitude(*scipy.linalg.eigh(white.T @ c0t @ white))
    proj = canonical(white @ u) * lam[None, :]
    return mean, proj[:, :dim]


def js_tica(ca_coords_dict, ref_key="target", n_bins=50, lagtime=20, return_tic=True, weights=None):
    """reference :169-200: TICA (2 components, fitted on the reference ensemble's pairwise distances) -> 50-bin histograms over the
    reference's range per component -> mean Jensen-Shannon distance.  -> results (, projections) like the reference."""
    w = _weights(weights, ca_coords_dict)
    ca_pwd = {k: pairwise_distance_ca(v) for k, v in ca_coords_dict.items()}
    try:
        from deeptime.decomposition import TICA

        tica = TICA(dim=2, lagtime=lagtime).fit(ca_pwd[ref_key]).fetch_model()
        ca_dr2d = {k: tica.transform(v) for k, v in ca_pwd.items()}
    except ImportError:
        mean, proj = tica_fit(ca_pwd[ref_key], lagtime, dim=2)
        ca_dr2d = {k: (v.astype(np.float64) - mean) @ proj for k, v in ca_pwd.items()}
    d_min, d_max = ca_dr2d[ref_key].min(axis=0), ca_dr2d[ref_key].max(axis=0)
    binned = {k: np.stack([np.histogram(v[:, c], bins=n_bins, weights=w[k], range=(d_min[c], d_max[c]))[0] + PSEUDO_C for c in range(v.shape[1])], 1)
              for k, v in ca_dr2d.items()}      # [n_bins, 2]
    results = {k: np.around(np.mean([_js(v[:, c], binned[ref_key][:, c]) for c in range(v.shape[1])]), decimals=4)
               for k, v in binned.items() if k != ref_key}
    results[ref_key] = 0.0
    if return_tic:
        return results, ca_dr2d
    return results
