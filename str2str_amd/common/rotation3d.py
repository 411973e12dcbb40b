"""Rotation conversions used at the host/boundary side of the sampler (torch, device-agnostic).

Same functions and semantics as the reference's ``src/common/rotation3d.py`` (:41-70, :102-161,
:461-553) but written branch-free with ``torch.where`` instead of boolean-mask indexing, so they
never force a device->host sync.  The hot loop does not call these: the per-step conversions live
in the fused HIP kernel (csrc/se3_step.hip, csrc/geom.h).
"""
from __future__ import annotations

import torch


def quaternion_to_matrix(quaternions: torch.Tensor) -> torch.Tensor:
    w, x, y, z = torch.unbind(quaternions, -1)
    two_s = 2.0 / (quaternions * quaternions).sum(-1)
    m = torch.stack(
        (
            1 - two_s * (y * y + z * z), two_s * (x * y - z * w), two_s * (x * z + y * w),
            two_s * (x * y + z * w), 1 - two_s * (x * x + z * z), two_s * (y * z - x * w),
            two_s * (x * z - y * w), two_s * (y * z + x * w), 1 - two_s * (x * x + y * y),
        ),
        -1,
    )
    return m.reshape(quaternions.shape[:-1] + (3, 3))


def matrix_to_quaternion(matrix: torch.Tensor) -> torch.Tensor:
    if matrix.size(-1) != 3 or matrix.size(-2) != 3:
        raise ValueError(f"Invalid rotation matrix shape {matrix.shape}.")
    lead = matrix.shape[:-2]
    m = matrix.reshape(lead + (9,))
    m00, m01, m02, m10, m11, m12, m20, m21, m22 = torch.unbind(m, dim=-1)
    sq = torch.stack(
        [1.0 + m00 + m11 + m22, 1.0 + m00 - m11 - m22, 1.0 - m00 + m11 - m22, 1.0 - m00 - m11 + m22], dim=-1
    )
    q_abs = torch.where(sq > 0, torch.sqrt(sq.clamp(min=0)), torch.zeros_like(sq))
    rows = torch.stack(
        [
            torch.stack([q_abs[..., 0] ** 2, m21 - m12, m02 - m20, m10 - m01], dim=-1),
            torch.stack([m21 - m12, q_abs[..., 1] ** 2, m10 + m01, m02 + m20], dim=-1),
            torch.stack([m02 - m20, m10 + m01, q_abs[..., 2] ** 2, m12 + m21], dim=-1),
            torch.stack([m10 - m01, m20 + m02, m21 + m12, q_abs[..., 3] ** 2], dim=-1),
        ],
        dim=-2,
    )
    rows = rows / (2.0 * q_abs[..., None].clamp(min=0.1))
    pick = q_abs.argmax(dim=-1)
    return torch.gather(rows, -2, pick[..., None, None].expand(lead + (1, 4))).squeeze(-2)


def _half_sinc(angles: torch.Tensor, half_angles: torch.Tensor) -> torch.Tensor:
    small = angles.abs() < 1e-6
    safe = torch.where(small, torch.ones_like(angles), angles)
    return torch.where(small, 0.5 - (angles * angles) / 48, torch.sin(half_angles) / safe)


def axis_angle_to_quaternion(axis_angle: torch.Tensor) -> torch.Tensor:
    angles = torch.norm(axis_angle, p=2, dim=-1, keepdim=True)
    half = angles * 0.5
    return torch.cat([torch.cos(half), axis_angle * _half_sinc(angles, half)], dim=-1)


def quaternion_to_axis_angle(quaternions: torch.Tensor) -> torch.Tensor:
    norms = torch.norm(quaternions[..., 1:], p=2, dim=-1, keepdim=True)
    half = torch.atan2(norms, quaternions[..., :1])
    return quaternions[..., 1:] / _half_sinc(2 * half, half)


def axis_angle_to_matrix(axis_angle: torch.Tensor) -> torch.Tensor:
    return quaternion_to_matrix(axis_angle_to_quaternion(axis_angle))


def matrix_to_axis_angle(matrix: torch.Tensor) -> torch.Tensor:
    return quaternion_to_axis_angle(matrix_to_quaternion(matrix))
