"""Seeded synthetic weights and inputs (no checkpoint or dataset ships with the reference).

Recipe per SURVEY.md §8(c)/(d): every tensor of the 274-key ``DenoisingNet.state_dict()`` is drawn
from one CPU ``torch.Generator`` in sorted-key order, with fan-in scaling for dense weights and a
small ``sigma_final`` for the layers the reference zero-initialises (``init="final"``,
src/models/net/layers.py:119-120: linear_out, skip_embed, trunk.linear_b, node_transition.linear_3,
bb_update, edge_transition.final_layer, torsion linear_3/linear_final).  sigma_final=0.02 gives an
ill-conditioned denoiser (per-op / teacher-forced checks); 0.002 gives a contractive sampler for
free-running trajectory parity.  The same recipe is loaded into the imported reference when the
golden fixtures are generated (tests/golden/make_golden.py), so no weights are committed.
"""
from __future__ import annotations

import math
from typing import Dict, Iterable, Tuple

import torch

_FINAL_MARKERS = (
    ".linear_out.",
    ".skip_embed_",
    ".trunk.linear_",
    ".linear_3.",
    ".bb_update_",
    ".final_layer.",
    ".linear_final.",
)


def _is_final(key: str) -> bool:
    k = "." + key
    return any(m in k for m in _FINAL_MARKERS)


def synth_state_dict(manifest: Iterable[Tuple[str, Tuple[int, ...]]], seed: int = 0,
                     sigma_final: float = 0.02, style: str = "init", dense_gain: float = 2.0) -> Dict[str, torch.Tensor]:
    """``style="trained_like"`` keeps the same draws but gives them the magnitudes a trained checkpoint typically has and a fresh
    initialisation does not: LayerNorm gains log-normal in [0.1, 10], dense weights ``dense_gain`` x the fan-in scale, biases
    0.1 -- hidden activations of several hundred to a few thousand (the regime that matters for the f16x3 kernels' range)."""
    if style not in ("init", "trained_like"):
        raise ValueError(style)
    tl = style == "trained_like"
    g = torch.Generator().manual_seed(int(seed))
    out = {}
    for key, shape in sorted((k, tuple(s)) for k, s in manifest):
        r = torch.randn(shape, generator=g, dtype=torch.float32)
        if key.endswith("head_weights"):
            v = 0.541324854612918 + 0.1 * r
        elif len(shape) == 2:
            v = r * (sigma_final if _is_final(key) else math.sqrt(1.0 / shape[1]) * (dense_gain if tl else 1.0))
        elif key.endswith(".weight"):  # 1-D weight == LayerNorm gain
            v = torch.exp(0.9 * r).clamp(0.1, 10.0) if tl else 1.0 + 0.05 * r
        else:  # biases (Linear and LayerNorm)
            v = r * (sigma_final if _is_final(key) else (0.1 if tl else 0.02))
        out[key] = v.contiguous()
    return out


def synth_chain(n_res: int, *, frame_seed: int = 3, aatype_seed: int = 4) -> Dict[str, torch.Tensor]:
    """The synthetic N-residue target of SURVEY.md §8(d): helix-like CA trace
    ca_k = (2.3 cos(1.745 k), 2.3 sin(1.745 k), 1.5 k) centred, seeded unit-quaternion frames,
    uniform aatype in 0..19, all masks 1, fixed_mask 0.  Returns the batch-of-1 feature dict the
    sampler reads (diffusion_module.py:249-257,271,340), dtypes as the data pipeline produces
    them (masks float64, frames float32 4x4)."""
    k = torch.arange(n_res, dtype=torch.float64)
    ca = torch.stack([2.3 * torch.cos(1.745 * k), 2.3 * torch.sin(1.745 * k), 1.5 * k], dim=-1)
    ca = (ca - ca.mean(dim=0, keepdim=True)).float()
    g = torch.Generator().manual_seed(frame_seed)
    q = torch.randn(n_res, 4, generator=g)
    q = q / q.norm(dim=-1, keepdim=True)
    a, b, c, d = q.unbind(-1)
    R = torch.stack(
        [
            torch.stack([a * a + b * b - c * c - d * d, 2 * b * c - 2 * a * d, 2 * b * d + 2 * a * c], -1),
            torch.stack([2 * b * c + 2 * a * d, a * a - b * b + c * c - d * d, 2 * c * d - 2 * a * b], -1),
            torch.stack([2 * b * d - 2 * a * c, 2 * c * d + 2 * a * b, a * a - b * b - c * c + d * d], -1),
        ],
        dim=-2,
    )
    frames = torch.zeros(1, n_res, 8, 4, 4)
    frames[0, :, 0, :3, :3] = R
    frames[0, :, 0, :3, 3] = ca
    frames[0, :, 0, 3, 3] = 1.0
    g2 = torch.Generator().manual_seed(aatype_seed)
    aatype = torch.randint(0, 20, (1, n_res), generator=g2, dtype=torch.int64)
    tors = torch.zeros(1, n_res, 7, 2, dtype=torch.float64)
    tors[..., 1] = 1.0
    idx = torch.arange(n_res, dtype=torch.int64)[None]
    return {
        "aatype": aatype,
        "residue_mask": torch.ones(1, n_res, dtype=torch.float64),
        "fixed_mask": torch.zeros(1, n_res, dtype=torch.float64),
        "residue_idx": idx,
        "residue_index": idx.clone(),
        "chain_index": torch.zeros(1, n_res, dtype=torch.int64),
        "torsion_angles_sin_cos": tors,
        "rigidgroups_gt_frames": frames,
        "accession_code": [f"synth{n_res}"],
    }
