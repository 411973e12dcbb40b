"""Oracle: SE(3) diffusion (IGSO(3) x VP-SDE) score / reverse / forward marginal and the
forward_backward sampler (TEST INFRASTRUCTURE — see oracle/__init__.py).

dtype behaviour is mirrored on purpose: the data pipeline hands float64 masks
(src/data/components/dataset.py:19-23,72-80), which promotes the scores and the R^3 update to
float64 (frame.py:136-138, r3.py:111-124); compose_rotvec is float64 by construction
(so3.py:18-19); Rigid casts back to float32 (rigid_utils.py:329-331,902).
"""
from __future__ import annotations

import math
from typing import Dict, Optional

import numpy as np
import torch

from . import geometry as G
from .geometry import Frames


def inflate(t: torch.Tensor, target: torch.Tensor) -> torch.Tensor:
    """src/utils/tensor_utils.py:24-43."""
    d = target.ndim - t.ndim
    return t[(...,) + (None,) * d] if d > 0 else t


# ------------------------------------------------------------------ so3.py
def compose_rotvec(r1: torch.Tensor, r2: torch.Tensor) -> torch.Tensor:
    """so3.py:13-19: R(r1) @ R(r2) in float64, back to r1's dtype."""
    R1 = G.axis_angle_to_matrix(r1)
    R2 = G.axis_angle_to_matrix(r2)
    cR = torch.einsum("...ij,...jk->...ik", R1.double(), R2.double())
    return G.matrix_to_axis_angle(cR).type(r1.dtype)


def igso3_expansion_np(omega: np.ndarray, eps, L: int = 1000) -> np.ndarray:
    """so3.py:21-62, numpy branch with 1-D omega (cache construction)."""
    ls = np.arange(L)[None]
    omega = omega[..., None]
    p = (2 * ls + 1) * np.exp(-ls * (ls + 1) * eps**2 / 2) * np.sin(omega * (ls + 1 / 2)) / np.sin(omega / 2)
    return p.sum(axis=-1)


def igso3_expansion_t(omega: torch.Tensor, eps: torch.Tensor, L: int = 1000) -> torch.Tensor:
    """so3.py:21-62, torch branch with omega [B,N], eps [B,1]."""
    ls = torch.arange(L)[None, None]
    omega = omega[..., None]
    eps = eps[..., None]
    p = (2 * ls + 1) * torch.exp(-ls * (ls + 1) * eps**2 / 2) * torch.sin(omega * (ls + 1 / 2)) / torch.sin(omega / 2)
    return p.sum(dim=-1)


def igso3_score_t(exp: torch.Tensor, omega: torch.Tensor, eps: torch.Tensor, L: int = 1000) -> torch.Tensor:
    """so3.py:85-130, torch branch: d/d omega log f via the quotient rule, / (f + 1e-4)."""
    ls = torch.arange(L)[None, None]
    omega = omega[..., None]
    eps = eps[..., None]
    hi = torch.sin(omega * (ls + 1 / 2))
    dhi = (ls + 1 / 2) * torch.cos(omega * (ls + 1 / 2))
    lo = torch.sin(omega / 2)
    dlo = 1 / 2 * torch.cos(omega / 2)
    dS = (2 * ls + 1) * torch.exp(-ls * (ls + 1) * eps**2 / 2) * (lo * dhi - hi * dlo) / lo**2
    return dS.sum(dim=-1) / (exp + 1e-4)


class SO3:
    """so3.py:133-370 minus the 1000x1000 disk cache: cdf rows are computed lazily with the same
    formulae (so3.py:185-187, :65-82)."""

    def __init__(self, min_sigma=0.1, max_sigma=1.5, num_sigma=1000, num_omega=1000, eps=1e-6):
        self.min_sigma, self.max_sigma = min_sigma, max_sigma
        self.num_sigma, self.num_omega, self.eps = num_sigma, num_omega, eps
        self.discrete_omega = torch.linspace(0, np.pi, steps=num_omega + 1)[1:]
        self._cdf_rows: Dict[int, np.ndarray] = {}

    @property
    def discrete_sigma(self):
        return self.sigma(torch.linspace(0.0, 1.0, self.num_sigma))

    def sigma(self, t: torch.Tensor):  # so3.py:216-223
        return torch.log(t * math.exp(self.max_sigma) + (1 - t) * math.exp(self.min_sigma))

    def sigma_idx(self, sigma: torch.Tensor):  # so3.py:211-214
        return torch.as_tensor(np.digitize(sigma.cpu().numpy(), self.discrete_sigma) - 1, dtype=torch.long)

    def t_to_idx(self, t):  # so3.py:236-238
        return self.sigma_idx(self.sigma(t))

    def diffusion_coef(self, t):  # so3.py:225-234
        return torch.sqrt(
            2 * (math.exp(self.max_sigma) - math.exp(self.min_sigma)) * self.sigma(t) / torch.exp(self.sigma(t))
        )

    def cdf_row(self, idx: int) -> np.ndarray:  # so3.py:176-187 for one sigma
        if idx not in self._cdf_rows:
            om = self.discrete_omega.numpy()
            sg = self.discrete_sigma.numpy()[idx]
            ex = igso3_expansion_np(om, sg)
            pdf = ex * (1.0 - np.cos(om)) / np.pi
            self._cdf_rows[idx] = pdf.cumsum() / self.num_omega * np.pi
        return self._cdf_rows[idx]

    def sample(self, t: torch.Tensor, shape):  # so3.py:244-272
        z = torch.randn(shape)
        x = z / torch.linalg.norm(z, dim=-1, keepdims=True)
        u = torch.rand(shape[:-1])
        scal = []
        for i, _t in enumerate(t):
            idx = self.t_to_idx(_t).item()
            scal.append(np.interp(u[i], self.cdf_row(idx), self.discrete_omega))
        scal = torch.as_tensor(np.asarray(scal), dtype=x.dtype)
        return x * scal[..., None]

    def score(self, vec: torch.Tensor, t: torch.Tensor):  # so3.py:274-309 (use_cached_score=False)
        omega = torch.linalg.norm(vec, dim=-1) + self.eps
        sigma = torch.as_tensor(self.discrete_sigma[self.t_to_idx(t)])
        f = igso3_expansion_t(omega, sigma[:, None])
        s = igso3_score_t(f, omega, sigma[:, None])
        return s[..., None] * vec / (omega[..., None] + self.eps)

    def forward_marginal(self, rot_0, t):  # so3.py:315-331
        d = self.sample(t, rot_0.shape)
        return compose_rotvec(rot_0, d), self.score(d, t)

    def reverse(self, rot_t, score_t, t, dt, noise_scale=1.0, probability_flow=True):  # so3.py:333-370
        t = inflate(t, rot_t)
        g = self.diffusion_coef(t)
        z = noise_scale * torch.randn_like(score_t)
        drift = -1.0 * (g**2) * score_t * dt * (0.5 if probability_flow else 1.0)
        diff = 0.0 if probability_flow else (g * np.sqrt(dt) * z)
        return compose_rotvec(rot_t, -1.0 * (drift + diff))


# ------------------------------------------------------------------ r3.py
class R3:
    def __init__(self, min_b=0.1, max_b=20.0, coordinate_scaling=0.1):
        self.min_b, self.max_b, self.cs = min_b, max_b, coordinate_scaling

    def b_t(self, t):  # r3.py:26-29
        return self.min_b + t * (self.max_b - self.min_b)

    def marginal_b_t(self, t):  # r3.py:40-41
        return t * self.min_b + 0.5 * (t**2) * (self.max_b - self.min_b)

    def conditional_var(self, t):  # r3.py:127-131
        return 1.0 - torch.exp(-self.marginal_b_t(t))

    def score(self, x_t, x_0, t, scale=False):  # r3.py:133-137
        t = inflate(t, x_t)
        if scale:
            x_t, x_0 = x_t * self.cs, x_0 * self.cs
        return -(x_t - torch.exp(-0.5 * self.marginal_b_t(t)) * x_0) / self.conditional_var(t)

    def forward_marginal(self, x_0, t):  # r3.py:49-74
        t = inflate(t, x_0)
        x_0 = x_0 * self.cs
        loc = torch.exp(-0.5 * self.marginal_b_t(t)) * x_0
        scale = torch.sqrt(1 - torch.exp(-self.marginal_b_t(t)))
        z = torch.randn_like(x_0)
        x_t = z * scale + loc
        return x_t / self.cs, self.score(x_t, x_0, t)

    def reverse(self, x_t, score_t, t, dt, center=True, noise_scale=1.0, probability_flow=True):  # r3.py:79-125
        t = inflate(t, x_t)
        x_t = x_t * self.cs
        f = -0.5 * self.b_t(t) * x_t
        g = torch.sqrt(self.b_t(t))
        z = noise_scale * torch.randn_like(score_t)
        drift = (f - g**2 * score_t) * dt * (0.5 if probability_flow else 1.0)
        diff = 0.0 if probability_flow else (g * math.sqrt(dt) * z)
        mask = torch.ones_like(x_t[..., 0])
        x1 = x_t - (drift + diff)
        if center:
            com = torch.sum(x1, dim=-2) / torch.sum(mask, dim=-1)[..., None]
            x1 = x1 - com[..., None, :]
        return x1 / self.cs


# ------------------------------------------------------------------ frame.py
def _assemble(rotvec, trans) -> Frames:  # frame.py:9-15
    return Frames(trans, rot_mats=G.axis_angle_to_matrix(rotvec))


def _apply_mask(tgt, src, m):  # frame.py:17-18
    return m * tgt + (1 - m) * src


class FrameDiffuser:
    def __init__(self, r3: Optional[R3] = None, so3: Optional[SO3] = None, min_t=0.01):
        self.r3 = r3 or R3()
        self.so3 = so3 or SO3()
        self.min_t = min_t

    def forward_marginal(self, rigids_0: Frames, t, diffuse_mask=None):  # frame.py:36-107
        rot_0 = G.matrix_to_axis_angle(rigids_0.get_rot_mats())
        trans_0 = rigids_0.trans
        rot_t, _ = self.so3.forward_marginal(rot_0, t)
        trans_t, _ = self.r3.forward_marginal(trans_0, t)
        if diffuse_mask is not None:
            m = torch.as_tensor(diffuse_mask, dtype=trans_t.dtype)[..., None]
            rot_t = _apply_mask(rot_t, rot_0, m)
            trans_t = _apply_mask(trans_t, trans_0, m)
        return _assemble(rot_t, trans_t).to_tensor_7()

    def sample_prior(self, shape):  # frame.py:212-255 (no reference rigids)
        rot = self.so3.sample(torch.ones(shape[0], dtype=torch.float), tuple(shape) + (3,))
        trans = torch.randn(size=tuple(shape) + (3,))
        return _assemble(rot, trans / self.r3.cs).to_tensor_7()

    def score(self, x0: Frames, xt: Frames, t, mask=None):  # frame.py:109-143
        q0_inv = G.matrix_to_quaternion(
            Frames(x0.trans, quats=G.invert_quat(x0.quats)).get_rot_mats()
            if x0.quats is not None
            else x0.rot_mats.transpose(-1, -2)
        )
        qt = G.matrix_to_quaternion(xt.get_rot_mats())
        rotvec = G.quaternion_to_axis_angle(G.quat_multiply(q0_inv, qt))
        rot_score = self.so3.score(rotvec, t)
        trans_score = self.r3.score(xt.trans, x0.trans, t, scale=True)
        if mask is not None:
            trans_score = trans_score * mask[..., None]
            rot_score = rot_score * mask[..., None]
        return rot_score, trans_score

    def reverse(self, xt: Frames, rot_score, trans_score, t, dt, diffuse_mask=None, center=True,
                noise_scale=1.0, probability_flow=True) -> Frames:  # frame.py:153-210
        rot_t = G.matrix_to_axis_angle(xt.get_rot_mats())
        trans_t = xt.trans
        rot_1 = self.so3.reverse(rot_t, rot_score, t, dt, noise_scale, probability_flow)
        trans_1 = self.r3.reverse(trans_t, trans_score, t, dt, center, noise_scale, probability_flow)
        if diffuse_mask is not None:
            trans_1 = _apply_mask(trans_1, trans_t, diffuse_mask[..., None])
            rot_1 = _apply_mask(rot_1, rot_t, diffuse_mask[..., None])
        return _assemble(rot_1, trans_1)


# ------------------------------------------------------------------ diffusion_module.py:260-334
def forward_backward(net_fn, diffuser: FrameDiffuser, feats: dict, rigids_0: Frames, t_delta: float, *,
                     num_timesteps: int, min_t: float = 0.01, noise_scale: float = 1.0,
                     probability_flow: bool = True, self_conditioning: bool = True, trace: Optional[list] = None,
                     rigids_t_init: Optional[torch.Tensor] = None):
    """The sampler closure of DiffusionLitModule.predict_step (diffusion_module.py:260-334).

    ``feats`` holds the B-repeated aatype / residue_mask / fixed_mask / residue_idx /
    torsion_angles_sin_cos (:269-272).  ``net_fn(batch) -> dict(rigids=Frames, psi=...)``.
    Returns atom37 [B,N,37,3] float32 numpy.  If ``trace`` is a list, every step appends
    dict(t, rigids_t(in), x0(7), psi, rigids_next(7)).  ``rigids_t_init`` [B,N,7] replaces the forward-marginal draw (a fixture's
    noised frames: the loop alone).
    """
    from .geometry import compute_backbone

    T = t_delta if t_delta > 0 else 1.0
    B = rigids_0.trans.shape[0]
    n = int(float(num_timesteps) * T)
    dt = 1.0 / n
    ts = np.linspace(min_t, T, n)[::-1]
    f = dict(feats)
    if rigids_t_init is not None:
        rigids_t = rigids_t_init.clone()
    elif t_delta > 0:
        rigids_t = diffuser.forward_marginal(rigids_0, t_delta * torch.ones(B), diffuse_mask=f["residue_mask"])
    else:
        rigids_t = diffuser.sample_prior(rigids_0.trans.shape[:-1])
    f["rigids_t"] = rigids_t
    diffuse_mask = (1 - f["fixed_mask"]) * f["residue_mask"]
    with torch.no_grad():
        if self_conditioning:
            f["sc_ca_t"] = torch.zeros_like(rigids_t[..., 4:])
            f["t"] = ts[0] * torch.ones(B)
            f["sc_ca_t"] = net_fn(f)["rigids"].to_tensor_7()[..., 4:]
        for t in ts:
            f["t"] = t * torch.ones(B)
            out = net_fn(f)
            if t == min_t:
                pred = out["rigids"]
                if trace is not None:
                    trace.append(dict(t=t, rigids_t=f["rigids_t"].clone(), x0=out["rigids"].to_tensor_7(),
                                      psi=out["psi"].clone(), rigids_next=None))
            else:
                x0_7 = out["rigids"].to_tensor_7()
                if self_conditioning:
                    f["sc_ca_t"] = x0_7[..., 4:]
                xt = Frames.from_tensor_7(f["rigids_t"])
                rs, tsc = diffuser.score(out["rigids"], xt, f["t"], mask=f["residue_mask"])
                pred = diffuser.reverse(xt, rs, tsc, f["t"], dt, diffuse_mask, True, noise_scale, probability_flow)
                nxt = pred.to_tensor_7()
                if trace is not None:
                    trace.append(dict(t=t, rigids_t=f["rigids_t"].clone(), x0=x0_7, psi=out["psi"].clone(),
                                      rot_score=rs, trans_score=tsc, rigids_next=nxt.clone()))
                f["rigids_t"] = nxt
        atom37 = compute_backbone(pred, out["psi"], f["aatype"])[0]
    return atom37.numpy()
