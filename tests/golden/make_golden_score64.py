"""float64 anchors for the IGSO(3) rotation score (authoring container only; needs /root/reference).

    python tests/golden/make_golden_score64.py     ->  tests/golden/score64.npz

The reference evaluates the 1000-term series of so3.py:21-62 / :85-130 in float32, and where the sum is tiny against its
terms its own result is rounding noise.  To judge another implementation fairly the parity tests need to know HOW noisy
the reference value is, residue by residue.  This script re-runs, on the inputs of the committed score fixtures
(score_reverse.npz, so3_score.npz, traj_teacher_n16.npz), the reference's OWN functions ``igso3_expansion`` and ``score``
with every argument promoted to float64 (the rotation vector is the one the reference's float32 conversion chain of
frame.py:121-128 produces, promoted) and stores that value next to the float32 one recomputed here (asserted equal to
the committed fixture).  The tests then require   |hip - ref64| <= |ref32 - ref64| + 4e-5 |s|   per residue.
"""
import numpy as np
import torch

import make_golden as G  # noqa: E402  (sets up the reference import shim)
from src.common import rotation3d  # noqa: E402
from src.common.rigid_utils import Rigid, quat_multiply  # noqa: E402
from src.models.score import so3 as ref_so3  # noqa: E402


def rotvec_0t(x0_7, xt_7):
    """frame.py:121-128 on tensor_7 inputs (float32 chain of the reference)."""
    r0, rt = Rigid.from_tensor_7(torch.as_tensor(x0_7)), Rigid.from_tensor_7(torch.as_tensor(xt_7))
    q0i = rotation3d.matrix_to_quaternion(r0.get_rots().invert().get_rot_mats())
    qt = rotation3d.matrix_to_quaternion(rt.get_rots().get_rot_mats())
    return rotation3d.quaternion_to_axis_angle(quat_multiply(q0i, qt))


def score_both(sd, vec32, t):
    """-> (reference float32 score, the same formula in float64) for rotation vectors vec32 [B,N,3], t [B]."""
    s32 = sd.score(vec32, t)
    sigma = sd.discrete_sigma[sd.t_to_idx(t)].double()
    vec = vec32.double()
    omega = torch.linalg.norm(vec, dim=-1) + sd.eps
    f = ref_so3.igso3_expansion(omega, sigma[:, None], use_torch=True)
    s = ref_so3.score(f, omega, sigma[:, None], use_torch=True)
    assert s.dtype == torch.float64 and f.dtype == torch.float64
    return s32, s[..., None] * vec / (omega[..., None] + sd.eps)


def main():
    diff = G.build_diffuser()
    sd = diff.rot_diffuser
    out = {}
    g = np.load(G.os.path.join(G.HERE, "score_reverse.npz"))
    t = torch.as_tensor(g["t"])
    v = rotvec_0t(g["x0"], g["xt"])
    s32, s64 = score_both(sd, v, t)
    m = torch.as_tensor(g["mask"])[..., None]
    assert np.array_equal((s32 * m).numpy(), g["rot_score"])
    out.update(sr_rotvec=v, sr_score64=s64 * m)

    g = np.load(G.os.path.join(G.HERE, "so3_score.npz"))
    s32, s64 = score_both(sd, torch.as_tensor(g["vec"]), torch.as_tensor(g["t"]))
    assert np.array_equal(s32.numpy(), g["score"])
    out.update(grid_score64=s64)

    g = np.load(G.os.path.join(G.HERE, "traj_teacher_n16.npz"))
    B = int(g["B"])
    vs, ss = [], []
    for i in range(len(g["ts"]) - 1):
        t = float(g["ts"][i]) * torch.ones(B)
        v = rotvec_0t(g["x0"][i], g["rigids_t"][i])
        s32, s64 = score_both(sd, v, t)
        assert np.array_equal(s32.double().numpy(), g["rot_score"][i]), i
        vs.append(v.numpy()); ss.append(s64.numpy())
    out.update(tf_rotvec=np.stack(vs), tf_score64=np.stack(ss))
    G.npz("score64.npz", **out)


if __name__ == "__main__":
    main()
