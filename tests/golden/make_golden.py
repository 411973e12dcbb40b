"""Generate the committed golden fixtures by RUNNING THE REFERENCE (authoring container only).

    python tests/golden/make_golden.py            # writes tests/golden/*.npz, *.json

The reference (read-only at /root/reference) is imported through tests/golden/_ref_import.py and
executed on CPU with seeded synthetic weights (str2str_amd/synth.py).  Only inputs and outputs are
stored; no reference source travels.  The sampler fixture re-types the control flow of
DiffusionLitModule.predict_step's ``forward_backward`` closure (diffusion_module.py:260-334)
around the imported net / diffuser / compute_backbone, because diffusion_module.py itself needs
lightning/torchmetrics, which are not installed.
"""
import json
import os
import sys
from copy import deepcopy

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)
import _ref_import  # noqa: E402

_ref_import.install()
import warnings  # noqa: E402

warnings.filterwarnings("ignore")

from src.common import all_atom, rotation3d  # noqa: E402
from src.common import rigid_utils as ru  # noqa: E402
from src.common.rigid_utils import Rigid, Rotation  # noqa: E402
from src.models.net.denoising_ipa import DenoisingNet, EmbeddingModule  # noqa: E402
from src.models.net.ipa import TranslationIPA  # noqa: E402
from src.models.score import frame as ref_frame  # noqa: E402
from src.models.score import r3 as ref_r3  # noqa: E402
from src.models.score import so3 as ref_so3  # noqa: E402

from str2str_amd.synth import synth_chain, synth_state_dict  # noqa: E402

torch.set_num_threads(8)
CACHE = os.environ.get("STR2STR_SO3_CACHE", "/tmp/str2str_so3_cache")


def npz(name, **arrs):
    out = {}
    for k, v in arrs.items():
        if isinstance(v, torch.Tensor):
            v = v.detach().cpu().numpy()
        out[k] = np.asarray(v)
    path = os.path.join(HERE, name)
    np.savez_compressed(path, **out)
    print(f"{name}: {os.path.getsize(path)/1024:.1f} KiB")


def build_net(seed=0, sigma_final=0.02):
    net = DenoisingNet(
        EmbeddingModule(32, 256, 128, 22, 1e-5, 20.0, True),
        TranslationIPA(256, 128, 0.1, 4, 64, 4, 2, 256, 8, 8, 12, 0.0),
    )
    manifest = [(k, tuple(v.shape)) for k, v in net.state_dict().items()]
    net.load_state_dict(synth_state_dict(manifest, seed=seed, sigma_final=sigma_final), strict=True)
    net.eval()
    return net, manifest


def build_diffuser():
    return ref_frame.FrameDiffuser(
        trans_diffuser=ref_r3.R3Diffuser(min_b=0.1, max_b=20.0, coordinate_scaling=0.1),
        rot_diffuser=ref_so3.SO3Diffuser(cache_dir=CACHE, num_omega=1000, num_sigma=1000, min_sigma=0.1,
                                         max_sigma=1.5, schedule="logarithmic", use_cached_score=False),
        min_t=1e-2,
    )


def rand_quats(g, *shape):
    q = torch.randn(*shape, 4, generator=g)
    return q / q.norm(dim=-1, keepdim=True)


def rand_rigids7(g, B, N, scale=10.0):
    return torch.cat([rand_quats(g, B, N), scale * torch.randn(B, N, 3, generator=g)], dim=-1)


# ------------------------------------------------------------------------------------------
def gen_prims():
    g = torch.Generator().manual_seed(11)
    q = rand_quats(g, 64)
    # edge cases: identity, near-identity, w<0, near-pi rotations, exact axis rotations
    extra = torch.tensor(
        [[1, 0, 0, 0], [-1, 0, 0, 0], [0, 1, 0, 0], [0, 0, 1, 0], [0, 0, 0, 1], [0.5, 0.5, 0.5, 0.5],
         [1, 1e-7, 0, 0], [1, 1e-4, -1e-4, 2e-4], [1e-4, 1, 0, 0], [-1e-4, 0.6, 0.8, 0], [0.70710678, 0.70710678, 0, 0]],
        dtype=torch.float32,
    )
    extra = extra / extra.norm(dim=-1, keepdim=True)
    q = torch.cat([q, extra], 0)
    R = ru.quat_to_rot(q)
    m2q = rotation3d.matrix_to_quaternion(R)
    q2aa = rotation3d.quaternion_to_axis_angle(q)
    aa = torch.cat([q2aa, torch.tensor([[0, 0, 0], [1e-7, 0, 0], [3e-7, -2e-7, 1e-7], [3.14159, 0, 0], [0, 4.0, 0]],
                                       dtype=torch.float32)], 0)
    aa2m = rotation3d.axis_angle_to_matrix(aa)
    aa2q = rotation3d.axis_angle_to_quaternion(aa)
    m2aa = rotation3d.matrix_to_axis_angle(R)
    q2 = rand_quats(g, q.shape[0])
    qmul = ru.quat_multiply(q, q2)
    vec = torch.randn(q.shape[0], 3, generator=g)
    qvec = ru.quat_multiply_by_vec(q, vec)
    q2m = rotation3d.quaternion_to_matrix(q)
    # compose_q_update_vec on a Rigid (quats) with partial mask
    t = torch.randn(q.shape[0], 3, generator=g)
    upd = 0.3 * torch.randn(q.shape[0], 6, generator=g)
    msk = (torch.rand(q.shape[0], 1, generator=g) > 0.2).float()
    rig = Rigid(Rotation(quats=q, normalize_quats=False), t).compose_q_update_vec(upd, msk)
    # compose_rotvec (float64 island)
    aa_b = torch.randn(aa.shape[0], 3, generator=g).double() * 0.1
    crv = ref_so3.compose_rotvec(aa, aa_b)
    # from_3_points
    p = torch.randn(32, 3, 3, generator=g)
    f3 = Rigid.from_3_points(p[:, 0], p[:, 1], p[:, 2])
    npz("prims.npz", q=q, R=R, m2q=m2q, q2aa=q2aa, aa=aa, aa2m=aa2m, aa2q=aa2q, m2aa=m2aa, q2=q2, qmul=qmul,
        vec=vec, qvec=qvec, q2m=q2m, t=t, upd=upd, msk=msk, comp7=rig.to_tensor_7(), aa_b=aa_b, crv=crv,
        p3=p, f3_rot=f3.get_rots().get_rot_mats(), f3_trans=f3.get_trans())


def gen_backbone():
    g = torch.Generator().manual_seed(12)
    aatype = torch.arange(21)[None].repeat(2, 1)
    aatype[1] = torch.randint(0, 21, (21,), generator=g)
    r7 = rand_rigids7(g, 2, 21)
    psi = torch.randn(2, 21, 2, generator=g)
    psi = (psi / psi.norm(dim=-1, keepdim=True)).double()
    a37, m37, _, a14 = all_atom.compute_backbone(Rigid.from_tensor_7(r7), psi, aatype)
    npz("backbone.npz", aatype=aatype, rigids7=r7, psi=psi, atom37=a37, mask37=m37, atom14=a14)


def gen_modules(net):
    g = torch.Generator().manual_seed(13)
    B, N = 2, 16
    tr = net.translator.trunk
    s = torch.randn(B, N, 256, generator=g)
    z = torch.randn(B, N, N, 128, generator=g)
    r7 = rand_rigids7(g, B, N, scale=1.0)  # translations already in nm (x0.1) scale
    mask = torch.ones(B, N)
    mask[1, -3:] = 0
    with torch.no_grad():
        out = tr["ipa_0"](s, z, Rigid.from_tensor_7(r7), mask)
        et = tr["edge_transition_0"](s, z)
        nt = tr["node_transition_0"](s)
        tor = net.translator.torsion_pred(s)
        x = torch.randn(N, B, 320, generator=g)
        tro = tr["transformer_0"](x, src_key_padding_mask=1.0 - mask)
    npz("ipa.npz", s=s, z=z, rigids7=r7, mask=mask, out=out)
    npz("edge_transition.npz", node=s, edge=z, out=et)
    npz("node_modules.npz", s=s, node_transition=nt, torsion=tor, x=x, mask=mask, transformer=tro)

    # embedding incl. diagonal (d=0) and exact bin-edge distances
    idx = torch.arange(N)[None].repeat(B, 1)
    idx[1] = idx[1] * 2 + 5  # gaps / offset numbering
    t = torch.tensor([0.37, 0.01])
    fixed = torch.zeros(B, N)
    fixed[1, :2] = 1
    ca = 6.0 * torch.randn(B, N, 3, generator=g)
    lower = torch.linspace(1e-5, 20.0, 22)
    ca[0, 1] = ca[0, 0] + torch.tensor([float(lower[3]), 0.0, 0.0])  # on a bin edge
    ca[0, 2] = ca[0, 0] + torch.tensor([0.0, 25.0, 0.0])  # beyond last edge
    ca[1, 1] = ca[1, 0]  # zero distance off-diagonal
    with torch.no_grad():
        node, edge = net.embedder(residue_idx=idx, t=t, fixed_mask=fixed, self_conditioning_ca=ca)
    npz("embedding.npz", residue_idx=idx, t=t, fixed_mask=fixed, sc_ca=ca, node=node, edge=edge)


def make_batch(g, B, N, partial_mask=False):
    feats = synth_chain(N)
    batch = {k: v.repeat(B, *(1,) * (v.ndim - 1)) for k, v in feats.items()
             if k in ("aatype", "residue_mask", "fixed_mask", "residue_idx", "torsion_angles_sin_cos")}
    if partial_mask:
        batch["residue_mask"][-1, -2:] = 0
    batch["rigids_t"] = rand_rigids7(g, B, N, scale=8.0)
    batch["sc_ca_t"] = 8.0 * torch.randn(B, N, 3, generator=g)
    batch["t"] = torch.rand(B, generator=g) * 0.9 + 0.05
    return batch


def gen_net(net):
    g = torch.Generator().manual_seed(14)
    for tag, (B, N, pm) in {"b1n10": (1, 10, False), "b2n16": (2, 16, True)}.items():
        batch = make_batch(g, B, N, pm)
        with torch.no_grad():
            out = net(batch)
        npz(f"net_{tag}.npz", **{f"in_{k}": v for k, v in batch.items()}, rigids7=out["rigids"].to_tensor_7(),
            psi=out["psi"], atom37=out["atom37"][..., :5, :], atom14=out["atom14"][..., :5, :])


def gen_diffuser(diff):
    g = torch.Generator().manual_seed(15)
    B, N = 4, 12
    t = torch.tensor([0.01, 0.1, 0.5, 1.0])
    x0 = rand_rigids7(g, B, N, scale=8.0)
    # x_t: a perturbed x0 (small for small t) and one unrelated sample
    dq = torch.cat([torch.ones(B, N, 1), torch.randn(B, N, 3, generator=g) * t[:, None, None]], -1)
    qt = ru.quat_multiply(x0[..., :4], dq / dq.norm(dim=-1, keepdim=True))
    xt = torch.cat([qt, x0[..., 4:] + 3.0 * t[:, None, None] * torch.randn(B, N, 3, generator=g)], -1)
    # rigids_t as the loop feeds it: through matrix -> quaternion
    xt = Rigid(Rotation(rot_mats=ru.quat_to_rot(xt[..., :4])), xt[..., 4:]).to_tensor_7()
    mask = torch.ones(B, N, dtype=torch.float64)
    mask[2, -2:] = 0
    x0r = Rigid(Rotation(quats=x0[..., :4], normalize_quats=True), x0[..., 4:])
    sc = diff.score(rigids_0=x0r, rigids_t=Rigid.from_tensor_7(xt), t=t, mask=mask)
    torch.manual_seed(99)
    dt = 1.0 / 100
    nxt = diff.reverse(rigids_t=Rigid.from_tensor_7(xt), rot_score=sc["rot_score"], trans_score=sc["trans_score"], t=t,
                       dt=dt, diffuse_mask=mask, center_trans=True, noise_scale=1.0, probability_flow=True)
    torch.manual_seed(99)
    nxt_sde = diff.reverse(rigids_t=Rigid.from_tensor_7(xt), rot_score=sc["rot_score"], trans_score=sc["trans_score"],
                           t=t, dt=dt, diffuse_mask=mask, center_trans=True, noise_scale=1.0, probability_flow=False)
    npz("score_reverse.npz", t=t, x0=x0r.to_tensor_7(), xt=xt, mask=mask, rot_score=sc["rot_score"],
        trans_score=sc["trans_score"], dt=dt, next7=nxt.to_tensor_7(), next7_sde=nxt_sde.to_tensor_7())

    # SO(3) score over a grid of omega for several t (so3.py:274-309)
    tt = torch.tensor([0.01, 0.05, 0.1, 0.5, 1.0])
    om = torch.linspace(1e-4, 2 * np.pi - 1e-3, 48)
    axis = torch.randn(48, 3, generator=g)
    axis = axis / axis.norm(dim=-1, keepdim=True)
    vec = (om[:, None] * axis)[None].repeat(5, 1, 1)
    s = diff.rot_diffuser.score(vec, tt)
    sd = diff.rot_diffuser
    npz("so3_score.npz", t=tt, vec=vec, score=s, sigma=sd.sigma(tt), sigma_idx=sd.t_to_idx(tt),
        discrete_sigma=sd.discrete_sigma, g=sd.diffusion_coef(tt), cdf_rows=sd._cdf[[0, 70, 500, 999]].numpy(),
        cdf_row_idx=np.array([0, 70, 500, 999]), score_scaling=sd._score_scaling.numpy())

    # forward marginal / prior with the global CPU generator
    gt = Rigid.from_tensor_7(rand_rigids7(g, 3, 10, scale=8.0))
    gt4 = gt.to_tensor_4x4()
    m = torch.ones(3, 10, dtype=torch.float64)
    m[0, :2] = 0
    torch.manual_seed(123)
    fm = diff.forward_marginal(rigids_0=Rigid.from_tensor_4x4(gt4), t=0.35 * torch.ones(3), diffuse_mask=m,
                               as_tensor_7=True)["rigids_t"]
    torch.manual_seed(124)
    pr = diff.sample_prior(shape=torch.Size([3, 10]), device="cpu", as_tensor_7=True)["rigids_t"]
    npz("forward_marginal.npz", gt4=gt4, mask=m, t_delta=0.35, seed_fm=123, rigids_t=fm, seed_prior=124, prior=pr)


def ref_forward_backward(net, diff, batch, rigids_0, t_delta, *, num_timesteps, min_t=0.01, noise_scale=1.0,
                         probability_flow=True, self_conditioning=True, trace=None):
    """Control flow of diffusion_module.py:260-334 around the imported reference objects."""
    T = t_delta if t_delta > 0 else 1.0
    batch_size, device = rigids_0.shape[0], rigids_0.device
    _n = int(float(num_timesteps) * T)
    dt = 1.0 / _n
    ts = np.linspace(min_t, T, _n)[::-1]
    _feats = deepcopy({k: v.repeat(batch_size, *(1,) * (v.ndim - 1)) for k, v in batch.items()
                       if k in ("aatype", "residue_mask", "fixed_mask", "residue_idx", "torsion_angles_sin_cos")})
    if t_delta > 0:
        rigids_t = diff.forward_marginal(rigids_0=rigids_0, t=t_delta * torch.ones(batch_size, device=device),
                                         diffuse_mask=_feats["residue_mask"], as_tensor_7=True)["rigids_t"]
    else:
        rigids_t = diff.sample_prior(shape=rigids_0.shape, device=device, as_tensor_7=True)["rigids_t"]
    _feats["rigids_t"] = rigids_t
    with torch.no_grad():
        diffuse_mask = (1 - _feats["fixed_mask"]) * _feats["residue_mask"]
        if self_conditioning:
            _feats["sc_ca_t"] = torch.zeros_like(rigids_t[..., 4:])
            _feats["t"] = ts[0] * torch.ones(batch_size, device=device)
            _feats["sc_ca_t"] = net(_feats, as_tensor_7=True)["rigids"][..., 4:]
        for t in ts:
            _feats["t"] = t * torch.ones(batch_size, device=device)
            out = net(_feats, as_tensor_7=False)
            if t == min_t:
                rigids_pred = out["rigids"]
                if trace is not None:
                    trace.append(dict(t=t, rigids_t=_feats["rigids_t"].clone(), sc_ca_t=_feats["sc_ca_t"].clone(),
                                      x0=out["rigids"].to_tensor_7(), psi=out["psi"].clone()))
            else:
                sc_in = _feats["sc_ca_t"].clone()
                if self_conditioning:
                    _feats["sc_ca_t"] = out["rigids"].to_tensor_7()[..., 4:]
                ps = diff.score(rigids_0=out["rigids"], rigids_t=Rigid.from_tensor_7(_feats["rigids_t"]), t=_feats["t"],
                                mask=_feats["residue_mask"])
                rigids_pred = diff.reverse(rigids_t=Rigid.from_tensor_7(_feats["rigids_t"]), rot_score=ps["rot_score"],
                                           trans_score=ps["trans_score"], t=_feats["t"], dt=dt, diffuse_mask=diffuse_mask,
                                           center_trans=True, noise_scale=noise_scale, probability_flow=probability_flow)
                nxt = rigids_pred.to_tensor_7()
                if trace is not None:
                    trace.append(dict(t=t, rigids_t=_feats["rigids_t"].clone(), sc_ca_t=sc_in,
                                      x0=out["rigids"].to_tensor_7(), psi=out["psi"].clone(),
                                      rot_score=ps["rot_score"], trans_score=ps["trans_score"], next7=nxt.clone()))
                _feats["rigids_t"] = nxt
        atom37 = all_atom.compute_backbone(rigids_pred, out["psi"], aatype=_feats["aatype"])[0]
    return atom37.detach().cpu().numpy(), ts, dt


def gen_trajectories(diff):
    # (1) teacher-forced material: every step's inputs and outputs, ill-conditioned weights
    net, _ = build_net(seed=0, sigma_final=0.02)
    feats = synth_chain(16)
    B = 2
    gt4 = feats["rigidgroups_gt_frames"][..., 0, :, :].clone()
    rig0 = Rigid.from_tensor_4x4(gt4.repeat(B, 1, 1, 1))
    trace = []
    torch.manual_seed(42)
    atom37, ts, dt = ref_forward_backward(net, diff, feats, rig0, 1.0, num_timesteps=20, trace=trace)
    keys = ["rigids_t", "sc_ca_t", "x0", "psi"]
    arr = {k: np.stack([s[k].numpy() for s in trace]) for k in keys}
    for k in ["rot_score", "trans_score", "next7"]:
        arr[k] = np.stack([s[k].numpy() for s in trace[:-1]])
    npz("traj_teacher_n16.npz", ts=ts.copy(), dt=dt, atom37=atom37[..., :5, :], seed=42, n_res=16, B=B, **arr)

    # (2) free-running, contractive weights (sigma_final = 0.002): final coordinates only
    net2, _ = build_net(seed=0, sigma_final=0.002)
    for tag, (N, B, S, td) in {"n16_s20": (16, 2, 20, 1.0), "cfg1_n64_s20": (64, 4, 20, 1.0),
                               "n24_delta": (24, 2, 40, 0.5), "n12_prior": (12, 2, 10, -1.0)}.items():
        feats = synth_chain(N)
        gt4 = feats["rigidgroups_gt_frames"][..., 0, :, :].clone()
        rig0 = Rigid.from_tensor_4x4(gt4.repeat(B, 1, 1, 1))
        torch.manual_seed(42)
        trace = []
        atom37, ts, dt = ref_forward_backward(net2, diff, feats, rig0, td, num_timesteps=S, trace=trace)
        npz(f"traj_free_{tag}.npz", atom37=atom37[..., :5, :], ts=ts.copy(), dt=dt, seed=42, n_res=N, B=B,
            num_timesteps=S, t_delta=td, first_rigids_t=trace[0]["rigids_t"], last_x0=trace[-1]["x0"])


def gen_schedule(diff):
    out = {}
    sd = diff.rot_diffuser
    r3 = diff.trans_diffuser
    for i, (nt, T) in enumerate([(20, 1.0), (100, 1.0), (1000, 0.25), (1000, 0.7)]):
        n = int(float(nt) * T)
        ts = np.linspace(0.01, T, n)[::-1].copy()
        tt = torch.stack([t * torch.ones(1) for t in ts])[:, 0]
        out[f"ts_{i}"] = ts
        out[f"dt_{i}"] = np.float64(1.0 / n)
        out[f"sigma_idx_{i}"] = sd.t_to_idx(tt).numpy()
        out[f"sigma_{i}"] = sd.discrete_sigma[sd.t_to_idx(tt)].numpy()
        out[f"g_rot_{i}"] = sd.diffusion_coef(tt).numpy()
        out[f"b_t_{i}"] = r3.b_t(tt).numpy()
        out[f"mb_t_{i}"] = r3.marginal_b_t(tt).numpy()
        out[f"cond_var_{i}"] = r3.conditional_var(tt).numpy()
    npz("schedule.npz", **out)


def main():
    net, manifest = build_net(seed=0, sigma_final=0.02)
    with open(os.path.join(HERE, "state_dict_manifest.json"), "w") as f:
        json.dump([[k, list(s)] for k, s in manifest], f, indent=0)
    diff = build_diffuser()
    gen_prims()
    gen_backbone()
    gen_modules(net)
    gen_net(net)
    gen_diffuser(diff)
    gen_schedule(diff)
    gen_trajectories(diff)


if __name__ == "__main__":
    main()
