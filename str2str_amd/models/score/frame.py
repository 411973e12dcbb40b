"""``FrameDiffuser``: SE(3) diffusion wrapper of the sampling path.

Same public methods as the reference's ``src/models/score/frame.py`` (forward_marginal :36-107,
score :109-143, reverse :153-210, sample_prior :212-255).  ``score`` and ``reverse`` enqueue the fused
HIP step (``s2s_se3_step``, csrc/se3_step.hip); ``step`` does both in ONE launch and is what the
sampler uses.  ``forward_marginal`` / ``sample_prior`` run once per trajectory on host tensors with
the global host generator, in the reference's draw order, so a fixed seed reproduces its noise.
"""
from __future__ import annotations

from typing import Optional

import numpy as np
import torch

from ... import ops
from ...common import rotation3d
from ...common.rigid_utils import Rigid, Rotation
from . import r3, so3


def assemble_rigid(rotvec: torch.Tensor, trans: torch.Tensor) -> Rigid:
    return Rigid(Rotation(rot_mats=rotation3d.axis_angle_to_matrix(rotvec)), trans)


def apply_mask(x_tgt, x_src, tgt_mask):
    return tgt_mask * x_tgt + (1 - tgt_mask) * x_src


def _as7(r) -> torch.Tensor:
    return r if isinstance(r, torch.Tensor) else r.to_tensor_7()


class FrameDiffuser:
    def __init__(self, trans_diffuser: Optional[r3.R3Diffuser] = None, rot_diffuser: Optional[so3.SO3Diffuser] = None,
                 min_t: float = 0.001):
        if trans_diffuser is None or rot_diffuser is None:
            raise NotImplementedError("the sampling path diffuses both translations and rotations")
        self.trans_diffuser, self.rot_diffuser, self.min_t = trans_diffuser, rot_diffuser, min_t

    # ------------------------------------------------------------------ once per trajectory (host)
    def forward_marginal(self, rigids_0: Rigid, t: torch.Tensor, diffuse_mask: torch.Tensor = None,
                         as_tensor_7: bool = True):
        dev = rigids_0.device
        r0 = rigids_0.to(device="cpu")
        t_h = t.detach().float().cpu()
        rot_0 = rotation3d.matrix_to_axis_angle(r0.get_rots().get_rot_mats())
        trans_0 = r0.get_trans()
        rot_t, _ = self.rot_diffuser.forward_marginal(rot_0, t_h)
        trans_t, _ = self.trans_diffuser.forward_marginal(trans_0, t_h)
        if diffuse_mask is not None:
            m = torch.as_tensor(diffuse_mask, dtype=trans_t.dtype).cpu()[..., None]
            rot_t = apply_mask(rot_t, rot_0, m)
            trans_t = apply_mask(trans_t, trans_0, m)
        rigids_t = assemble_rigid(rot_t, trans_t)
        out = rigids_t.to_tensor_7().to(dev) if as_tensor_7 else rigids_t.to(device=dev)
        return {"rigids_t": out}

    def sample_prior(self, shape, device=None, reference_rigids: Rigid = None, diffuse_mask: torch.Tensor = None,
                     as_tensor_7: bool = False):
        if reference_rigids is not None or diffuse_mask is not None:
            # (the reference's own motif branch asserts reference_rigids.shape[:-1] == shape, frame.py:224 -- i.e. it only accepts a sample
            #  shape WITHOUT the residue axis and then broadcasts ONE prior draw over all residues of a sample; predict_step never takes it)
            raise NotImplementedError("motif-conditioned priors are outside the sampling path")
        shape = tuple(shape)
        rot = self.rot_diffuser.sample_prior(shape=shape + (3,))
        trans = self.trans_diffuser.unscale(self.trans_diffuser.sample_prior(shape=shape + (3,)))
        rigids_t = assemble_rigid(rot, trans)
        return {"rigids_t": rigids_t.to_tensor_7().to(device) if as_tensor_7 else rigids_t.to(device=device)}

    def forward_marginal_device(self, rigids0_4x4: Optional[torch.Tensor], t_delta: Optional[float], diffuse_mask=None,
                                shape=None, noise=None) -> torch.Tensor:
        """Throughput-mode forward marginal (``rigids0_4x4`` [B,N,4,4] on the device, ``t_delta`` > 0) or prior sample
        (``rigids0_4x4=None``, ``shape=(B, N)``): the same arithmetic as ``forward_marginal`` / ``sample_prior``
        (reference frame.py:36-107, :212-255) in ONE launch (``s2s_forward_marginal``) on noise drawn by the device
        generator -- no per-replica host loop, no np.interp, no host->device copy.  ``noise`` = (z_axis [B,N,3], u [B,N],
        z_trans [B,N,3]) overrides the draws (parity test against the host path).  -> rigids_t as tensor_7 [B,N,7]."""
        prior = rigids0_4x4 is None
        dev = torch.device("cuda", torch.cuda.current_device()) if prior else rigids0_4x4.device
        B, N = tuple(shape) if prior else rigids0_4x4.shape[:2]
        t = torch.full((B,), 1.0 if prior else float(t_delta), dtype=torch.float32)
        sd = self.rot_diffuser
        idx = sd.t_to_idx(t)
        rows, inv = torch.unique(idx, return_inverse=True)
        cdf = torch.as_tensor(np.stack([sd.cdf_row(int(i)) for i in rows]), dtype=torch.float64).to(dev).contiguous()
        if noise is None:
            noise = (torch.randn(B, N, 3, device=dev), torch.rand(B, N, device=dev), torch.randn(B, N, 3, device=dev))
        z_axis, u, z_trans = (x.to(dev).float().contiguous() for x in noise)
        p2 = None
        if not prior:
            mb = self.trans_diffuser.marginal_b_t(t)
            p2 = torch.stack([torch.exp(-0.5 * mb), torch.sqrt(1 - torch.exp(-mb))], dim=-1).float().to(dev).contiguous()
            rigids0_4x4 = rigids0_4x4.float().contiguous()
        dm = None if diffuse_mask is None else diffuse_mask.to(dev).float().contiguous()
        return torch.ops.str2str_amd.forward_marginal(rigids0_4x4, z_axis, u, z_trans, cdf, inv.to(torch.int32).to(dev).contiguous(),
                                                      sd.discrete_omega.float().to(dev).contiguous(), p2, dm,
                                                      self.trans_diffuser.coordinate_scaling)

    # ------------------------------------------------------------------ per step (HIP)
    def step_params(self, t: torch.Tensor) -> torch.Tensor:
        """[B, 8] float32 host tensor (see include/str2str_hip.h, s2s_se3_step)."""
        sig, g2r, gr = self.rot_diffuser.step_params(t)
        eh, cv, bt, g2t, gt = self.trans_diffuser.step_params(t)
        return torch.stack([sig, g2r, eh, cv, bt, g2t, gr, gt], dim=-1).float().contiguous()

    def _masks(self, like: torch.Tensor, mask, diffuse_mask):
        B, N = like.shape[:2]
        one = None
        def cv(m):
            nonlocal one
            if m is None:
                one = torch.ones(B, N, device=like.device) if one is None else one
                return one
            return m.to(like.device).type(torch.float32).contiguous()
        return cv(mask), cv(diffuse_mask)

    def score(self, rigids_0, rigids_t, t: torch.Tensor, mask: torch.Tensor = None):
        x0, xt = _as7(rigids_0).float().contiguous(), _as7(rigids_t).float().contiguous()
        m, dm = self._masks(xt, mask, None)
        p8 = self.step_params(t).to(xt.device)
        _, rs, ts = ops.se3_step(x0, xt, m, dm, p8, dt=0.0, coordinate_scaling=self.trans_diffuser.coordinate_scaling,
                                 want_next=False, want_scores=True)
        return {"trans_score": ts, "rot_score": rs}

    def reverse(self, rigids_t, rot_score: torch.Tensor, trans_score: torch.Tensor, t: torch.Tensor, dt: float,
                diffuse_mask: torch.Tensor = None, center_trans: bool = True, noise_scale: float = 1.0,
                probability_flow: bool = True) -> Rigid:
        xt = _as7(rigids_t).float().contiguous()
        m, dm = self._masks(xt, None, diffuse_mask)
        p8 = self.step_params(t).to(xt.device)
        z_rot = z_trans = None
        if not probability_flow:  # host generator, reference order: rotation noise first, then translation
            z_rot = torch.randn(rot_score.shape, dtype=torch.float64).to(xt.device)
            z_trans = torch.randn(trans_score.shape, dtype=torch.float64).to(xt.device)
        nxt, _, _ = ops.se3_step(None, xt, m, dm, p8, dt=dt, coordinate_scaling=self.trans_diffuser.coordinate_scaling,
                                 probability_flow=probability_flow, center=center_trans, noise_scale=noise_scale,
                                 z_rot=z_rot, z_trans=z_trans, rot_score_in=rot_score.double().contiguous(),
                                 trans_score_in=trans_score.double().contiguous())
        return Rigid.from_tensor_7(nxt)

    def step(self, x0_7: torch.Tensor, xt_7: torch.Tensor, params8: torch.Tensor, dt: float, mask: torch.Tensor,
             diffuse_mask: torch.Tensor, center_trans: bool = True, noise_scale: float = 1.0,
             probability_flow: bool = True, z_rot=None, z_trans=None, want_scores: bool = False):
        """score + reverse + to_tensor_7 in one launch (diffusion_module.py:311-329 of the reference)."""
        return ops.se3_step(x0_7, xt_7, mask, diffuse_mask, params8, dt=dt,
                            coordinate_scaling=self.trans_diffuser.coordinate_scaling, probability_flow=probability_flow,
                            center=center_trans, noise_scale=noise_scale, z_rot=z_rot, z_trans=z_trans,
                            want_scores=want_scores)
