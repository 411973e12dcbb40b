"""``Rigid`` / ``Rotation`` value types of the sampling path (host-side mirror of the reference's
``src/common/rigid_utils.py`` surface that ``predict_step`` / ``FrameDiffuser`` / ``DenoisingNet``
pass around: SURVEY.md §8b "host-side types that must survive").

Storage is a float32 quaternion [*,4] OR a rotation matrix [*,3,3], plus a translation [*,3]
(reference: Rotation.__init__ :302-345 forces float32; Rigid.__init__ :860-905).  On HIP tensors
``compose_q_update_vec`` runs the fused kernel (csrc/rigid_kernels.hip); everything else here is
boundary code (a handful of calls per trajectory) in plain torch.
"""
from __future__ import annotations

from typing import Callable, Optional

import torch

from . import rotation3d


def quat_to_rot(quat: torch.Tensor) -> torch.Tensor:
    """Quadratic form without renormalisation (reference :187-207)."""
    a, b, c, d = torch.unbind(quat, -1)
    aa, bb, cc, dd = a * a, b * b, c * c, d * d
    m = torch.stack(
        [
            aa + bb - cc - dd, 2 * b * c - 2 * a * d, 2 * b * d + 2 * a * c,
            2 * b * c + 2 * a * d, aa - bb + cc - dd, 2 * c * d - 2 * a * b,
            2 * b * d - 2 * a * c, 2 * c * d + 2 * a * b, aa - bb - cc + dd,
        ],
        dim=-1,
    )
    return m.reshape(quat.shape[:-1] + (3, 3))


def quat_multiply(q1: torch.Tensor, q2: torch.Tensor) -> torch.Tensor:
    """Hamilton product (reference :256-265)."""
    a1, b1, c1, d1 = torch.unbind(q1, -1)
    a2, b2, c2, d2 = torch.unbind(q2, -1)
    return torch.stack(
        [
            a1 * a2 - b1 * b2 - c1 * c2 - d1 * d2,
            a1 * b2 + b1 * a2 + c1 * d2 - d1 * c2,
            a1 * c2 - b1 * d2 + c1 * a2 + d1 * b2,
            a1 * d2 + b1 * c2 - c1 * b2 + d1 * a2,
        ],
        dim=-1,
    )


def quat_multiply_by_vec(quat: torch.Tensor, vec: torch.Tensor) -> torch.Tensor:
    a, b, c, d = torch.unbind(quat, -1)
    x, y, z = torch.unbind(vec, -1)
    return torch.stack(
        [-b * x - c * y - d * z, a * x + c * z - d * y, a * y - b * z + d * x, a * z + b * y - c * x], dim=-1
    )


def rot_vec_mul(r: torch.Tensor, t: torch.Tensor) -> torch.Tensor:
    x, y, z = torch.unbind(t, -1)
    return torch.stack(
        [
            r[..., 0, 0] * x + r[..., 0, 1] * y + r[..., 0, 2] * z,
            r[..., 1, 0] * x + r[..., 1, 1] * y + r[..., 1, 2] * z,
            r[..., 2, 0] * x + r[..., 2, 1] * y + r[..., 2, 2] * z,
        ],
        dim=-1,
    )


def rot_matmul(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    cols = [rot_vec_mul(a, b[..., :, j]) for j in range(3)]
    return torch.stack(cols, dim=-1)


class Rotation:
    def __init__(self, rot_mats: Optional[torch.Tensor] = None, quats: Optional[torch.Tensor] = None,
                 normalize_quats: bool = True):
        if (rot_mats is None) == (quats is None):
            raise ValueError("Exactly one input argument must be specified")
        if (rot_mats is not None and rot_mats.shape[-2:] != (3, 3)) or (quats is not None and quats.shape[-1] != 4):
            raise ValueError("Incorrectly shaped rotation matrix or quaternion")
        if quats is not None:
            quats = quats.type(torch.float32)
            if normalize_quats:
                quats = quats / torch.linalg.norm(quats, dim=-1, keepdim=True)
        if rot_mats is not None:
            rot_mats = rot_mats.type(torch.float32)
        self._rot_mats = rot_mats
        self._quats = quats

    @property
    def shape(self) -> torch.Size:
        return self._rot_mats.shape[:-2] if self._rot_mats is not None else self._quats.shape[:-1]

    @property
    def device(self) -> torch.device:
        return (self._rot_mats if self._rot_mats is not None else self._quats).device

    def get_rot_mats(self) -> torch.Tensor:
        return self._rot_mats if self._rot_mats is not None else quat_to_rot(self._quats)

    def get_quats(self) -> torch.Tensor:
        return self._quats if self._quats is not None else rotation3d.matrix_to_quaternion(self._rot_mats)

    def get_cur_rot(self) -> torch.Tensor:
        return self._rot_mats if self._rot_mats is not None else self._quats

    def __getitem__(self, index) -> "Rotation":
        if not isinstance(index, tuple):
            index = (index,)
        if self._rot_mats is not None:
            return Rotation(rot_mats=self._rot_mats[index + (slice(None), slice(None))])
        return Rotation(quats=self._quats[index + (slice(None),)], normalize_quats=False)

    def invert(self) -> "Rotation":
        if self._rot_mats is not None:
            return Rotation(rot_mats=self._rot_mats.transpose(-1, -2))
        q = self._quats
        conj = torch.cat([q[..., :1], -q[..., 1:]], dim=-1)
        return Rotation(quats=conj / torch.sum(q**2, dim=-1, keepdim=True), normalize_quats=False)

    def apply(self, pts: torch.Tensor) -> torch.Tensor:
        return rot_vec_mul(self.get_rot_mats(), pts)

    def invert_apply(self, pts: torch.Tensor) -> torch.Tensor:
        return rot_vec_mul(self.get_rot_mats().transpose(-1, -2), pts)

    def compose_r(self, r: "Rotation") -> "Rotation":
        return Rotation(rot_mats=rot_matmul(self.get_rot_mats(), r.get_rot_mats()))

    def compose_q_update_vec(self, q_update_vec, normalize_quats: bool = True, update_mask=None) -> "Rotation":
        quats = self.get_quats()
        upd = quat_multiply_by_vec(quats, q_update_vec)
        if update_mask is not None:
            upd = upd * update_mask
        return Rotation(quats=quats + upd, normalize_quats=normalize_quats)

    def to(self, device=None, dtype=None) -> "Rotation":
        if self._rot_mats is not None:
            return Rotation(rot_mats=self._rot_mats.to(device=device))
        return Rotation(quats=self._quats.to(device=device), normalize_quats=False)


class Rigid:
    def __init__(self, rots: Optional[Rotation], trans: Optional[torch.Tensor]):
        if rots is None and trans is None:
            raise ValueError("At least one input argument must be specified")
        if rots is None:
            q = trans.new_zeros(trans.shape[:-1] + (4,), dtype=torch.float32)
            q[..., 0] = 1
            rots = Rotation(quats=q, normalize_quats=False)
        if trans is None:
            trans = torch.zeros(tuple(rots.shape) + (3,), dtype=torch.float32, device=rots.device)
        if rots.shape != trans.shape[:-1] or rots.device != trans.device:
            raise ValueError("Rots and trans incompatible")
        self._rots = rots
        self._trans = trans.type(torch.float32)

    @property
    def shape(self) -> torch.Size:
        return self._trans.shape[:-1]

    @property
    def device(self) -> torch.device:
        return self._trans.device

    def get_rots(self) -> Rotation:
        return self._rots

    def get_trans(self) -> torch.Tensor:
        return self._trans

    def __getitem__(self, index) -> "Rigid":
        if not isinstance(index, tuple):
            index = (index,)
        return Rigid(self._rots[index], self._trans[index + (slice(None),)])

    def to(self, device=None, dtype=None) -> "Rigid":
        return Rigid(self._rots.to(device=device), self._trans.to(device=device))

    # ---- tensor forms
    def to_tensor_7(self) -> torch.Tensor:
        return torch.cat([self._rots.get_quats(), self._trans], dim=-1)

    def to_tensor_4x4(self) -> torch.Tensor:
        t = self._trans.new_zeros(tuple(self.shape) + (4, 4))
        t[..., :3, :3] = self._rots.get_rot_mats()
        t[..., :3, 3] = self._trans
        t[..., 3, 3] = 1
        return t

    @staticmethod
    def from_tensor_7(t: torch.Tensor, normalize_quats: bool = False) -> "Rigid":
        if t.shape[-1] != 7:
            raise ValueError("Incorrectly shaped input tensor")
        return Rigid(Rotation(quats=t[..., :4], normalize_quats=normalize_quats), t[..., 4:])

    @staticmethod
    def from_tensor_4x4(t: torch.Tensor) -> "Rigid":
        if t.shape[-2:] != (4, 4):
            raise ValueError("Incorrectly shaped input tensor")
        return Rigid(Rotation(rot_mats=t[..., :3, :3]), t[..., :3, 3])

    @staticmethod
    def from_3_points(p_neg_x_axis, origin, p_xy_plane, eps: float = 1e-8) -> "Rigid":
        """Gram-Schmidt frame (AF2 algorithm 21; reference :1235-1278)."""
        e0 = origin - p_neg_x_axis
        e1 = p_xy_plane - origin
        e0 = e0 / torch.sqrt((e0 * e0).sum(-1, keepdim=True) + eps)
        e1 = e1 - e0 * (e0 * e1).sum(-1, keepdim=True)
        e1 = e1 / torch.sqrt((e1 * e1).sum(-1, keepdim=True) + eps)
        e2 = torch.cross(e0, e1, dim=-1)
        return Rigid(Rotation(rot_mats=torch.stack([e0, e1, e2], dim=-1)), origin)

    # ---- algebra
    def apply(self, pts: torch.Tensor) -> torch.Tensor:
        return self._rots.apply(pts) + self._trans

    def invert_apply(self, pts: torch.Tensor) -> torch.Tensor:
        return self._rots.invert_apply(pts - self._trans)

    def compose(self, r: "Rigid") -> "Rigid":
        return Rigid(self._rots.compose_r(r._rots), self._rots.apply(r._trans) + self._trans)

    def apply_trans_fn(self, fn: Callable[[torch.Tensor], torch.Tensor]) -> "Rigid":
        return Rigid(self._rots, fn(self._trans))

    def scale_translation(self, factor: float) -> "Rigid":
        return self.apply_trans_fn(lambda t: t * factor)

    def compose_q_update_vec(self, q_update_vec: torch.Tensor, update_mask: Optional[torch.Tensor] = None) -> "Rigid":
        """reference :1042-1066.  HIP tensors with a quaternion-format rotation use the fused kernel."""
        if q_update_vec.is_cuda and self._rots._quats is not None:
            from .. import ops

            lead = self.shape
            mask = (torch.ones(lead, device=self.device) if update_mask is None
                    else update_mask.reshape(lead).type(torch.float32)).contiguous()
            out = ops.rigid_compose_update(self.to_tensor_7().contiguous(), q_update_vec.type(torch.float32).contiguous(), mask)
            return Rigid(Rotation(quats=out[..., :4], normalize_quats=False), out[..., 4:])
        q_vec, t_vec = q_update_vec[..., :3], q_update_vec[..., 3:]
        new_rots = self._rots.compose_q_update_vec(q_vec, update_mask=update_mask)
        upd = self._rots.apply(t_vec)
        if update_mask is not None:
            upd = upd * update_mask
        return Rigid(new_rots, self._trans + upd)
