"""Oracle: rigid-frame / rotation primitives (TEST INFRASTRUCTURE — see oracle/__init__.py).

Plain torch restatement, CPU eager.  A rigid frame is carried as a pair of tensors: either
``("quat", q[...,4], t[...,3])`` or ``("mat", R[...,3,3], t[...,3])`` wrapped in the tiny
``Frames`` record below, mirroring the two storage formats of the reference ``Rigid``
(src/common/rigid_utils.py:302-345, 860-905).
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Optional

import torch


# ----------------------------------------------------------------------------- rotation3d.py
def quaternion_to_matrix(q: torch.Tensor) -> torch.Tensor:
    """src/common/rotation3d.py:41-70 (two_s = 2/|q|^2 form)."""
    r, i, j, k = torch.unbind(q, -1)
    two_s = 2.0 / (q * q).sum(-1)
    o = torch.stack(
        (
            1 - two_s * (j * j + k * k),
            two_s * (i * j - k * r),
            two_s * (i * k + j * r),
            two_s * (i * j + k * r),
            1 - two_s * (i * i + k * k),
            two_s * (j * k - i * r),
            two_s * (i * k - j * r),
            two_s * (j * k + i * r),
            1 - two_s * (i * i + j * j),
        ),
        -1,
    )
    return o.reshape(q.shape[:-1] + (3, 3))


def _sqrt_positive_part(x: torch.Tensor) -> torch.Tensor:
    """src/common/rotation3d.py:91-99."""
    return torch.where(x > 0, torch.sqrt(torch.clamp(x, min=0)), torch.zeros_like(x))


def matrix_to_quaternion(m: torch.Tensor) -> torch.Tensor:
    """src/common/rotation3d.py:102-161: four candidates, pick argmax |q| component (first max),
    denominators floored at 0.1, no sign standardisation."""
    batch = m.shape[:-2]
    m00, m01, m02, m10, m11, m12, m20, m21, m22 = torch.unbind(m.reshape(batch + (9,)), dim=-1)
    q_abs = _sqrt_positive_part(
        torch.stack(
            [1.0 + m00 + m11 + m22, 1.0 + m00 - m11 - m22, 1.0 - m00 + m11 - m22, 1.0 - m00 - m11 + m22],
            dim=-1,
        )
    )
    cand = torch.stack(
        [
            torch.stack([q_abs[..., 0] ** 2, m21 - m12, m02 - m20, m10 - m01], dim=-1),
            torch.stack([m21 - m12, q_abs[..., 1] ** 2, m10 + m01, m02 + m20], dim=-1),
            torch.stack([m02 - m20, m10 + m01, q_abs[..., 2] ** 2, m12 + m21], dim=-1),
            torch.stack([m10 - m01, m20 + m02, m21 + m12, q_abs[..., 3] ** 2], dim=-1),
        ],
        dim=-2,
    )
    flr = torch.tensor(0.1, dtype=q_abs.dtype)
    cand = cand / (2.0 * q_abs[..., None].max(flr))
    idx = q_abs.argmax(dim=-1)
    return torch.gather(cand, -2, idx[..., None, None].expand(batch + (1, 4))).squeeze(-2)


def axis_angle_to_quaternion(v: torch.Tensor) -> torch.Tensor:
    """src/common/rotation3d.py:493-522 (|angle| < 1e-6 -> 1/2 - angle^2/48)."""
    angles = torch.norm(v, p=2, dim=-1, keepdim=True)
    half = angles * 0.5
    small = angles.abs() < 1e-6
    safe = torch.where(small, torch.ones_like(angles), angles)
    s = torch.where(small, 0.5 - (angles * angles) / 48, torch.sin(half) / safe)
    return torch.cat([torch.cos(half), v * s], dim=-1)


def quaternion_to_axis_angle(q: torch.Tensor) -> torch.Tensor:
    """src/common/rotation3d.py:525-553 (half = atan2(|xyz|, w), angle = 2*half)."""
    norms = torch.norm(q[..., 1:], p=2, dim=-1, keepdim=True)
    half = torch.atan2(norms, q[..., :1])
    angles = 2 * half
    small = angles.abs() < 1e-6
    safe = torch.where(small, torch.ones_like(angles), angles)
    s = torch.where(small, 0.5 - (angles * angles) / 48, torch.sin(half) / safe)
    return q[..., 1:] / s


def axis_angle_to_matrix(v: torch.Tensor) -> torch.Tensor:
    """src/common/rotation3d.py:461-474."""
    return quaternion_to_matrix(axis_angle_to_quaternion(v))


def matrix_to_axis_angle(m: torch.Tensor) -> torch.Tensor:
    """src/common/rotation3d.py:477-490."""
    return quaternion_to_axis_angle(matrix_to_quaternion(m))


# ----------------------------------------------------------------------------- rigid_utils.py
def quat_to_rot(q: torch.Tensor) -> torch.Tensor:
    """src/common/rigid_utils.py:163-207: quadratic form, NO renormalisation."""
    a, b, c, d = torch.unbind(q, -1)
    rows = [
        [a * a + b * b - c * c - d * d, 2 * b * c - 2 * a * d, 2 * b * d + 2 * a * c],
        [2 * b * c + 2 * a * d, a * a - b * b + c * c - d * d, 2 * c * d - 2 * a * b],
        [2 * b * d - 2 * a * c, 2 * c * d + 2 * a * b, a * a - b * b - c * c + d * d],
    ]
    return torch.stack([torch.stack(r, dim=-1) for r in rows], dim=-2)


def quat_multiply(p: torch.Tensor, q: torch.Tensor) -> torch.Tensor:
    """src/common/rigid_utils.py:233-265 (Hamilton product)."""
    a1, b1, c1, d1 = torch.unbind(p, -1)
    a2, b2, c2, d2 = torch.unbind(q, -1)
    return torch.stack(
        [
            a1 * a2 - b1 * b2 - c1 * c2 - d1 * d2,
            a1 * b2 + b1 * a2 + c1 * d2 - d1 * c2,
            a1 * c2 - b1 * d2 + c1 * a2 + d1 * b2,
            a1 * d2 + b1 * c2 - c1 * b2 + d1 * a2,
        ],
        dim=-1,
    )


def quat_multiply_by_vec(q: torch.Tensor, v: torch.Tensor) -> torch.Tensor:
    """src/common/rigid_utils.py:268-277: q * (0, v)."""
    a, b, c, d = torch.unbind(q, -1)
    x, y, z = torch.unbind(v, -1)
    return torch.stack(
        [-b * x - c * y - d * z, a * x + c * z - d * y, a * y - b * z + d * x, a * z + b * y - c * x],
        dim=-1,
    )


def invert_quat(q: torch.Tensor) -> torch.Tensor:
    """src/common/rigid_utils.py:284-288."""
    qp = q.clone()
    qp[..., 1:] *= -1
    return qp / torch.sum(q**2, dim=-1, keepdim=True)


def rot_matmul(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    """src/common/rigid_utils.py:24-81 (hand-written 3x3 product, same term order)."""
    rows = []
    for i in range(3):
        rows.append(
            torch.stack(
                [
                    a[..., i, 0] * b[..., 0, j] + a[..., i, 1] * b[..., 1, j] + a[..., i, 2] * b[..., 2, j]
                    for j in range(3)
                ],
                dim=-1,
            )
        )
    return torch.stack(rows, dim=-2)


def rot_vec_mul(r: torch.Tensor, t: torch.Tensor) -> torch.Tensor:
    """src/common/rigid_utils.py:84-108."""
    x, y, z = t[..., 0], t[..., 1], t[..., 2]
    return torch.stack(
        [
            r[..., 0, 0] * x + r[..., 0, 1] * y + r[..., 0, 2] * z,
            r[..., 1, 0] * x + r[..., 1, 1] * y + r[..., 1, 2] * z,
            r[..., 2, 0] * x + r[..., 2, 1] * y + r[..., 2, 2] * z,
        ],
        dim=-1,
    )


@dataclass
class Frames:
    """Stand-in for the reference ``Rigid`` (rigid_utils.py:860-905): exactly one of quats /
    rot_mats is set; both are forced to float32 as the reference does (:329-331, :902)."""

    trans: torch.Tensor
    quats: Optional[torch.Tensor] = None
    rot_mats: Optional[torch.Tensor] = None

    def __post_init__(self):
        self.trans = self.trans.type(torch.float32)
        if self.quats is not None:
            self.quats = self.quats.type(torch.float32)
        if self.rot_mats is not None:
            self.rot_mats = self.rot_mats.type(torch.float32)

    # rigid_utils.py:509-523
    def get_rot_mats(self) -> torch.Tensor:
        return self.rot_mats if self.rot_mats is not None else quat_to_rot(self.quats)

    # rigid_utils.py:525-545 (matrix format goes through rotation3d.matrix_to_quaternion)
    def get_quats(self) -> torch.Tensor:
        return self.quats if self.quats is not None else matrix_to_quaternion(self.rot_mats)

    # rigid_utils.py:1203-1215
    def to_tensor_7(self) -> torch.Tensor:
        return torch.cat([self.get_quats(), self.trans], dim=-1)

    # rigid_utils.py:1217-1233 (normalize_quats=False)
    @staticmethod
    def from_tensor_7(t: torch.Tensor) -> "Frames":
        return Frames(trans=t[..., 4:], quats=t[..., :4])

    # rigid_utils.py:1182-1201
    @staticmethod
    def from_tensor_4x4(t: torch.Tensor) -> "Frames":
        return Frames(trans=t[..., :3, 3], rot_mats=t[..., :3, :3])

    # rigid_utils.py:1107-1120
    def apply(self, pts: torch.Tensor) -> torch.Tensor:
        return rot_vec_mul(self.get_rot_mats(), pts) + self.trans

    # rigid_utils.py:1122-1133
    def invert_apply(self, pts: torch.Tensor) -> torch.Tensor:
        return rot_vec_mul(self.get_rot_mats().transpose(-1, -2), pts - self.trans)

    def scale_translation(self, f: float) -> "Frames":
        """apply_trans_fn(x * f) (ipa.py:288-292)."""
        return Frames(self.trans * f, quats=self.quats, rot_mats=self.rot_mats)

    def unscale_translation(self, f: float) -> "Frames":
        return Frames(self.trans / f, quats=self.quats, rot_mats=self.rot_mats)

    # rigid_utils.py:1042-1066 + :590-619
    def compose_q_update_vec(self, upd: torch.Tensor, mask: torch.Tensor) -> "Frames":
        q_vec, t_vec = upd[..., :3], upd[..., 3:]
        quats = self.get_quats()
        new_q = quats + quat_multiply_by_vec(quats, q_vec) * mask
        new_q = new_q.type(torch.float32)
        new_q = new_q / torch.linalg.norm(new_q, dim=-1, keepdim=True)
        trans_update = rot_vec_mul(self.get_rot_mats(), t_vec) * mask
        return Frames(self.trans + trans_update, quats=new_q)


def from_3_points(p_neg_x_axis, origin, p_xy_plane, eps: float = 1e-8) -> Frames:
    """src/common/rigid_utils.py:1235-1278 (Gram-Schmidt, AF2 algorithm 21)."""
    e0 = origin - p_neg_x_axis
    e1 = p_xy_plane - origin
    e0 = e0 / torch.sqrt((e0 * e0).sum(-1, keepdim=True) + eps)
    dot = (e0 * e1).sum(-1, keepdim=True)
    e1 = e1 - e0 * dot
    e1 = e1 / torch.sqrt((e1 * e1).sum(-1, keepdim=True) + eps)
    e2 = torch.cross(e0, e1, dim=-1)
    rots = torch.stack([e0, e1, e2], dim=-1)
    return Frames(origin, rot_mats=rots)


# ----------------------------------------------------------------------------- all_atom.py
_TABLES = None


def _tables():
    global _TABLES
    if _TABLES is None:
        import os

        import numpy as np

        z = np.load(os.path.join(os.path.dirname(__file__), "backbone_tables.npz"))
        _TABLES = {k: torch.as_tensor(z[k]) for k in z.files}
    return _TABLES


def compute_backbone(frames: Frames, psi: torch.Tensor, aatype: Optional[torch.Tensor] = None):
    """src/common/all_atom.py:141-173 (+ :21-83, :99-138) restricted to what it outputs:
    atom14[..., :5] = N, CA, C, O, CB; atom37 slots 0..4 = N, CA, C, CB, O; everything else 0.

    group 0 (N, CA, C, CB): default frame = identity, torsion rotation (sin,cos) = (0,1).
    group 3 (O): default_frame[aatype, 3] composed with Rx(psi), then with the backbone frame.
    psi[..., 0] = sin, psi[..., 1] = cos; rotation [[1,0,0],[0,cos,-sin],[0,sin,cos]] (:44-57).
    """
    tb = _tables()
    shape = frames.trans.shape[:-1]
    if aatype is None:
        aatype = torch.zeros(shape, dtype=torch.long)
    aatype = aatype.long()
    pos = tb["pos"][aatype]  # [*, 5, 3]
    amask = tb["mask"][aatype]  # [*, 5]
    group = tb["group"][aatype]  # [*, 5]
    dflt = tb["frames"][aatype]  # [*, 2, 4, 4]
    d_rot, d_trans = dflt[..., :3, :3], dflt[..., :3, 3]  # [*,2,3,3], [*,2,3]

    # torsion rotations for groups 0 and 3 (cast to f32 by Rotation.__init__)
    sin = torch.stack([torch.zeros_like(psi[..., 0]), psi[..., 0]], dim=-1).type(torch.float32)
    cos = torch.stack([torch.ones_like(psi[..., 1]), psi[..., 1]], dim=-1).type(torch.float32)
    tor = torch.zeros(shape + (2, 3, 3), dtype=torch.float32)
    tor[..., 0, 0] = 1
    tor[..., 1, 1] = cos
    tor[..., 1, 2] = -sin
    tor[..., 2, 1] = sin
    tor[..., 2, 2] = cos
    # default_r.compose(all_rots): rot = Rd @ Rtor ; trans = Rd @ 0 + td
    g_rot = rot_matmul(d_rot, tor)
    g_trans = rot_vec_mul(d_rot, torch.zeros_like(d_trans)) + d_trans
    # r[..., None].compose(all_frames_to_bb)
    R = frames.get_rot_mats()[..., None, :, :]
    G_rot = rot_matmul(R, g_rot)
    G_trans = rot_vec_mul(R, g_trans) + frames.trans[..., None, :]
    # pick frame per atom (one-hot sum over groups in the reference == a gather)
    gi = (group == 3).long()  # index into our 2-group table
    A_rot = torch.gather(G_rot, -3, gi[..., None, None].expand(shape + (5, 3, 3)))
    A_trans = torch.gather(G_trans, -2, gi[..., None].expand(shape + (5, 3)))
    atom14 = (rot_vec_mul(A_rot, pos) + A_trans) * amask[..., None]
    atom37 = torch.zeros(shape + (37, 3), dtype=torch.float32)
    atom37[..., :3, :] = atom14[..., :3, :]
    atom37[..., 3, :] = atom14[..., 4, :]
    atom37[..., 4, :] = atom14[..., 3, :]
    atom14_full = torch.zeros(shape + (14, 3), dtype=torch.float32)
    atom14_full[..., :5, :] = atom14
    return atom37, torch.any(atom37 != 0, dim=-1), aatype, atom14_full
