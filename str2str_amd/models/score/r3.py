"""VP-SDE on translations: host-side schedule of the sampling path.

Constructor and method names follow the reference's ``src/models/score/r3.py``.  Only the scalar
schedule functions are evaluated here (float32 torch on the host, same formulae: b_t :26-29,
marginal_b_t :40-41, conditional_var :127-131); the per-residue score / reverse arithmetic
(:79-125, :133-137) runs inside the fused HIP step (csrc/se3_step.hip) and the once-per-trajectory
``forward_marginal`` (:49-74) is host tensor code.
"""
from __future__ import annotations

import torch


def _inflate(t, target: torch.Tensor):
    if isinstance(t, float):
        return t
    d = target.ndim - t.ndim
    return t[(...,) + (None,) * d] if d > 0 else t


class R3Diffuser:
    def __init__(self, min_b: float = 0.1, max_b: float = 20.0, coordinate_scaling: float = 1.0):
        self.min_b, self.max_b, self.coordinate_scaling = min_b, max_b, coordinate_scaling

    def scale(self, x):
        return x * self.coordinate_scaling

    def unscale(self, x):
        return x / self.coordinate_scaling

    def b_t(self, t: torch.Tensor):
        if torch.any(t < 0) or torch.any(t > 1):
            raise ValueError(f"Invalid t={t}")
        return self.min_b + t * (self.max_b - self.min_b)

    def diffusion_coef(self, t):
        return torch.sqrt(self.b_t(t))

    def drift_coef(self, x, t):
        return -0.5 * self.b_t(t) * x

    def marginal_b_t(self, t):
        return t * self.min_b + 0.5 * (t**2) * (self.max_b - self.min_b)

    def conditional_var(self, t, use_torch=False):
        return 1.0 - torch.exp(-self.marginal_b_t(t))

    def score_scaling(self, t: torch.Tensor):
        return 1.0 / torch.sqrt(self.conditional_var(t))

    def sample_prior(self, shape, device=None):
        return torch.randn(size=shape, device=device)

    def score(self, x_t, x_0, t, scale=False):
        """Host tensors only (used by forward_marginal); the sampler's score is in the HIP step."""
        t = _inflate(t, x_t)
        if scale:
            x_t, x_0 = self.scale(x_t), self.scale(x_0)
        return -(x_t - torch.exp(-0.5 * self.marginal_b_t(t)) * x_0) / self.conditional_var(t)

    def forward_marginal(self, x_0: torch.Tensor, t: torch.Tensor):
        t = _inflate(t, x_0)
        x_0 = self.scale(x_0)
        loc = torch.exp(-0.5 * self.marginal_b_t(t)) * x_0
        std = torch.sqrt(1 - torch.exp(-self.marginal_b_t(t)))
        z = torch.randn_like(x_0)
        x_t = z * std + loc
        return self.unscale(x_t), self.score(x_t, x_0, t)

    def step_params(self, t: torch.Tensor):
        """[B] float32 host tensors the fused step needs: exp(-beta/2), 1-exp(-beta), b(t), g^2, g."""
        t = t.detach().float().cpu()
        mb = self.marginal_b_t(t)
        g = self.diffusion_coef(t)
        return torch.exp(-0.5 * mb), self.conditional_var(t), self.b_t(t), g**2, g
