"""float64 anchors for the IGSO(3) rotation score (authoring container only; needs /root/reference).

    python tests/golden/make_golden_score64.py     ->  tests/golden/score64.npz

The reference evaluates the 1000-term series of so3.py:21-62 / :85-130 in float32, and where the sum is tiny against its
terms its own result is rounding noise.  To judge another implementation fairly the parity tests need to know HOW noisy
the reference value is, residue by residue.  This script re-runs, on the inputs of the committed score fixtures
(score_reverse.npz, so3_score.npz, traj_teacher_n16.npz), the reference's OWN functions ``igso3_expansion`` and ``score``
with every argument promoted to float64 (the rotation vector is the one the reference's float32 conversion chain of
frame.py:121-128 produces, promoted) and stores that value next to the float32 one recomputed here (asserted equal to
the committed fixture).  A second anchor promotes the WHOLE chain of frame.py:121-128 (invert_quat, quat_to_rot,
matrix_to_quaternion, quat_multiply, quaternion_to_axis_angle -- the reference's own functions, which keep the input dtype
once the float32-forcing Rigid/Rotation containers are bypassed) to float64: it measures how far the reference's float32
rotation vector itself is from exact arithmetic.  That matters where the un-standardised quaternion sign yields
omega = 2*pi - delta: omega is rebuilt as |xyz| * angle / sin(angle/2) with angle/2 one float32 ulp from pi, so a 1-ulp
difference between two libm atan2f / sinf implementations moves omega by (2*pi/delta) ulps.
A realised rounding error is one draw, not a bound, so the conditioning is also MEASURED with the reference itself:
the float32 reference is re-run K = 64 times with every input quaternion component moved by one float32 ulp in a random
direction (torch.nextafter); spread32 = the largest change of its own output.  Any float32 implementation of the same chain
(another libm, another summation order) is entitled to that much.
Input jitter rarely flips the rounding of the one operation that dominates the wrap case, angle/2 = atan2(|xyz|, w)
next to pi, so that term is added in closed form and evaluated with the reference's float64 series, by finite difference:
alg32 = |s64(v (1 +- r)) - s64(v)|,  r = K_LIBM * ulp32(omega/2) * |cot(omega/2)|  -- the forward error of
omega = |xyz| * angle / sin(angle/2) (rotation3d.py:525-553) when angle/2 is off by K_LIBM = 2 float32 ulps (atan2f and
sinf, one ulp each; torch's CPU kernels and the GPU's libm are different implementations).
The tests require, per residue, with s64 = series in float64 on the float32 rotation vector:
    |ours - s64| <= |ref32 - s64| + 2 * spread32 + alg32 + 4e-5 |s64|
(the reference's realised series noise + its measured 1-ulp sensitivity, twice because the distance between two
implementations is the difference of two such draws + the libm term), nothing excluded.  The bound itself is validated
against the reference on CPU (tests/test_oracle_golden.py): the reference's own realised chain error |c64 - s64|, with
c64 = the WHOLE chain in float64, must lie inside  2 * spread32 + alg32 + 4e-5 |s64|  on every fixture residue.
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


def rotvec_0t_f64(x0_7, xt_7):
    """The same chain on float64 tensors, through the reference's functions (Rigid / Rotation force float32, so the
    quaternion -> matrix steps are called directly: rigid_utils.py:187-207, :284-288; from_tensor_7 does not normalise)."""
    from src.common.rigid_utils import invert_quat, quat_to_rot

    q0 = torch.as_tensor(x0_7).double()[..., :4]
    qt = torch.as_tensor(xt_7).double()[..., :4]
    q0i = rotation3d.matrix_to_quaternion(quat_to_rot(invert_quat(q0)))
    qtm = rotation3d.matrix_to_quaternion(quat_to_rot(qt))
    v = rotation3d.quaternion_to_axis_angle(quat_multiply(q0i, qtm))
    assert v.dtype == torch.float64
    return v


def series64(sd, vec, t):
    sigma = sd.discrete_sigma[sd.t_to_idx(t)].double()
    vec = vec.double()
    omega = torch.linalg.norm(vec, dim=-1) + sd.eps
    f = ref_so3.igso3_expansion(omega, sigma[:, None], use_torch=True)
    s = ref_so3.score(f, omega, sigma[:, None], use_torch=True)
    assert s.dtype == torch.float64 and f.dtype == torch.float64
    return s[..., None] * vec / (omega[..., None] + sd.eps)


K_PERTURB = 64
K_LIBM = 2.0


def alg_term(sd, vec32, t):
    """|s64(v (1 +- r)) - s64(v)| with r = K_LIBM ulp32(omega/2) |cot(omega/2)| (see the module docstring)."""
    v = torch.as_tensor(vec32).double()
    om = torch.linalg.norm(v, dim=-1)
    ulp = torch.as_tensor(np.spacing((om / 2).float().numpy())).double()
    r = (K_LIBM * ulp / torch.tan(om / 2).abs().clamp(min=1e-300))[..., None]
    base = series64(sd, v, t)
    return torch.maximum((series64(sd, v * (1 + r), t) - base).norm(dim=-1), (series64(sd, v * (1 - r), t) - base).norm(dim=-1))



def ulp_jitter(x, gen):
    """Every element moved by exactly one float32 ulp, up or down at random."""
    x = torch.as_tensor(x).float()
    up = torch.rand(x.shape, generator=gen) < 0.5
    return torch.where(up, torch.nextafter(x, torch.full_like(x, float("inf"))), torch.nextafter(x, torch.full_like(x, -float("inf"))))


def spread_chain(sd, x0_7, xt_7, t, base32):
    """max_k |ref32(inputs + 1-ulp jitter_k) - ref32(inputs)| per residue, reference float32 code throughout."""
    gen = torch.Generator().manual_seed(2024)
    x0_7, xt_7 = torch.as_tensor(x0_7).float(), torch.as_tensor(xt_7).float()
    worst = torch.zeros(base32.shape[:-1], dtype=torch.float64)
    for _ in range(K_PERTURB):
        a, b = x0_7.clone(), xt_7.clone()
        a[..., :4], b[..., :4] = ulp_jitter(a[..., :4], gen), ulp_jitter(b[..., :4], gen)
        s = sd.score(rotvec_0t(a, b), t)
        worst = torch.maximum(worst, (s.double() - base32.double()).norm(dim=-1))
    return worst


def score_both(sd, vec32, t):
    """-> (reference float32 score, the same formula in float64) for rotation vectors vec32 [B,N,3], t [B]."""
    s32 = sd.score(vec32, t)
    return s32, series64(sd, vec32, t)


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
    out.update(sr_rotvec=v, sr_score64=s64 * m, sr_chain64=series64(sd, rotvec_0t_f64(g["x0"], g["xt"]), t) * m,
               sr_spread32=spread_chain(sd, g["x0"], g["xt"], t, s32) * m[..., 0], sr_alg32=alg_term(sd, v, t) * m[..., 0])

    g = np.load(G.os.path.join(G.HERE, "so3_score.npz"))
    s32, s64 = score_both(sd, torch.as_tensor(g["vec"]), torch.as_tensor(g["t"]))
    assert np.array_equal(s32.numpy(), g["score"])
    gen = torch.Generator().manual_seed(2025)
    worst = torch.zeros(s32.shape[:-1], dtype=torch.float64)
    for _ in range(K_PERTURB):  # the grid test feeds rotation vectors: jitter those
        sj = sd.score(ulp_jitter(g["vec"], gen), torch.as_tensor(g["t"]))
        worst = torch.maximum(worst, (sj.double() - s32.double()).norm(dim=-1))
    out.update(grid_score64=s64, grid_spread32=worst, grid_alg32=alg_term(sd, torch.as_tensor(g["vec"]), torch.as_tensor(g["t"])))

    g = np.load(G.os.path.join(G.HERE, "traj_teacher_n16.npz"))
    B = int(g["B"])
    vs, ss, cs, sp, al = [], [], [], [], []
    for i in range(len(g["ts"]) - 1):
        t = float(g["ts"][i]) * torch.ones(B)
        v = rotvec_0t(g["x0"][i], g["rigids_t"][i])
        s32, s64 = score_both(sd, v, t)
        assert np.array_equal(s32.double().numpy(), g["rot_score"][i]), i
        vs.append(v.numpy()); ss.append(s64.numpy())
        cs.append(series64(sd, rotvec_0t_f64(g["x0"][i], g["rigids_t"][i]), t).numpy())
        sp.append(spread_chain(sd, g["x0"][i], g["rigids_t"][i], t, s32).numpy())
        al.append(alg_term(sd, v, t).numpy())
    out.update(tf_rotvec=np.stack(vs), tf_score64=np.stack(ss), tf_chain64=np.stack(cs), tf_spread32=np.stack(sp),
               tf_alg32=np.stack(al))
    G.npz("score64.npz", **out)


if __name__ == "__main__":
    main()
