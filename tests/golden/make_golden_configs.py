"""Reference-generated fixtures for the BASELINE configs the N = 256 fixtures do not cover (authoring container only; RUNS THE
REFERENCE on CPU, stores inputs / outputs only):

    python tests/golden/make_golden_configs.py --cfg3      # configs[2]: the 12 Science2011 targets, 3 replicas, 10 + 1 evaluations each
    python tests/golden/make_golden_configs.py --cfg4      # configs[3]: N = 512 -- one evaluation + a free-running 10-step trajectory
    python tests/golden/make_golden_configs.py --cfg5      # configs[4]: per-chain UN-PADDED reference trajectories of mixed lengths
    python tests/golden/make_golden_configs.py --trained   # one evaluation with trained-like weight magnitudes (+ the reference's own
                                                           # thread-count noise as the yardstick)

cfg3 goes through the reference's own ProteinFeatureTransform (src/data/components/dataset.py) on the bundled PDB files, then the
control flow of DiffusionLitModule.predict_step (diffusion_module.py:260-351: one chunk of 3 replicas, t_delta = 0.5,
num_timesteps = 20) around the imported net / diffuser -- make_golden.ref_forward_backward."""
import os
import sys
import types

import numpy as np
import torch

import make_golden as G  # noqa: E402  (sets up the reference import shim)

HERE = G.HERE
CODES = ["1FME", "2F4K", "2JOF", "2WAV", "A3D", "CLN025", "GTT", "NTL9", "NuG2", "PRB", "UVF", "lambda"]


def _ref_transform():
    for name in ["Bio", "Bio.PDB", "biotite", "biotite.structure", "biotite.structure.io", "biotite.structure.io.pdb",
                 "lightning", "hydra", "hydra.utils"]:
        if name not in sys.modules:
            m = types.ModuleType(name)
            m.__path__ = []
            sys.modules[name] = m
    sys.modules["Bio.PDB"].PDBParser = object
    sys.modules["biotite.structure.io.pdb"].PDBFile = object
    sys.modules["lightning"].LightningDataModule = object
    sys.modules["hydra.utils"].instantiate = lambda *a, **k: None
    sys.modules["biotite.structure"].io = sys.modules["biotite.structure.io"]
    sys.modules["biotite"].structure = sys.modules["biotite.structure"]
    from src.data.components.dataset import ProteinFeatureTransform as RefTransform

    return RefTransform(strip_missing_residues=False, recenter_and_scale=False)


def cfg3():
    from str2str_amd.common import protein as my_protein   # the reference's reader needs Bio.PDB (see make_golden_io.py)

    tf = _ref_transform()
    diff = G.build_diffuser()
    net, _ = G.build_net(seed=0, sigma_final=0.002)
    out = {}
    for k, code in enumerate(CODES):
        with open(os.path.join(HERE, "pdb", f"{code}.pdb")) as f:
            feats = tf(my_protein.from_pdb_string(f.read()).to_dict())
        batch = {kk: feats[kk][None] for kk in ("aatype", "residue_mask", "fixed_mask", "residue_idx", "torsion_angles_sin_cos")}
        B, S, td = 3, 20, 0.5
        rig0 = G.Rigid.from_tensor_4x4(feats["rigidgroups_gt_frames"][None, :, 0].clone().repeat(B, 1, 1, 1))
        torch.manual_seed(100 + k)
        trace = []
        atom37, ts, dt = G.ref_forward_backward(net, diff, batch, rig0, td, num_timesteps=S, trace=trace)
        out[f"{code}/atom37"] = atom37[..., :5, :]
        out[f"{code}/first_rigids_t"] = trace[0]["rigids_t"].numpy()
        out[f"{code}/seed"] = 100 + k
        print(code, atom37.shape, flush=True)
    G.npz("cfg3_science2011.npz", B=3, num_timesteps=20, t_delta=0.5, **out)


def cfg4():
    from str2str_amd.synth import synth_chain

    net, _ = G.build_net(seed=0, sigma_final=0.02)
    g = torch.Generator().manual_seed(512)
    batch = G.make_batch(g, 1, 512, False)
    with torch.no_grad():
        out = net(batch)
    G.npz("net_b1n512.npz", **{f"in_{k}": v for k, v in batch.items()}, rigids7=out["rigids"].to_tensor_7(), psi=out["psi"],
          atom37=out["atom37"][..., :5, :], atom14=out["atom14"][..., :5, :])
    diff = G.build_diffuser()
    net2, _ = G.build_net(seed=0, sigma_final=0.002)
    N, B, S, td = 512, 1, 10, 1.0
    feats = synth_chain(N)
    rig0 = G.Rigid.from_tensor_4x4(feats["rigidgroups_gt_frames"][..., 0, :, :].clone().repeat(B, 1, 1, 1))
    torch.manual_seed(512)
    trace = []
    atom37, ts, dt = G.ref_forward_backward(net2, diff, feats, rig0, td, num_timesteps=S, trace=trace)
    G.npz("traj_free_n512_s10.npz", atom37=atom37[..., :5, :], ts=ts.copy(), dt=dt, seed=512, n_res=N, B=B, num_timesteps=S,
          t_delta=td, first_rigids_t=trace[0]["rigids_t"], last_x0=trace[-1]["x0"],
          **{f"rigids_t_step{k}": trace[k]["rigids_t"] for k in (5, 9)})


CFG5_LENS = (12, 33, 71, 214, 323)   # two short chains + three of the seed-5 draw U[64, 384] of the cfg5 workload


def cfg5():
    """Each chain ALONE and un-padded through the reference (R = 2 replicas, 4 denoise steps + the self-conditioning evaluation):
    the padded, masked multi-chain batch of this build must reproduce every one of them."""
    from str2str_amd.synth import synth_chain

    diff = G.build_diffuser()
    net2, _ = G.build_net(seed=0, sigma_final=0.002)
    R, S, td = 2, 4, 1.0
    out = {}
    for n in CFG5_LENS:
        feats = synth_chain(n, frame_seed=3 + n, aatype_seed=4 + n)
        rig0 = G.Rigid.from_tensor_4x4(feats["rigidgroups_gt_frames"][..., 0, :, :].clone().repeat(R, 1, 1, 1))
        torch.manual_seed(700 + n)
        trace = []
        atom37, ts, dt = G.ref_forward_backward(net2, diff, feats, rig0, td, num_timesteps=S, trace=trace)
        out[f"n{n}/atom37"] = atom37[..., :5, :]
        out[f"n{n}/first_rigids_t"] = trace[0]["rigids_t"].numpy()
        print(n, atom37.shape, flush=True)
    G.npz("cfg5_mixed_lengths.npz", lens=np.array(CFG5_LENS), R=R, num_timesteps=S, t_delta=td, **out)


def trained(B=2, N=24, seed=77, name="net_b2n24_trained_like.npz"):
    """One evaluation (B = 2, N = 24, partial mask; --trained256: B = 1, N = 256, the bench shape -> net_b1n256_trained_like.npz)
    with the trained-like weight recipe of str2str_amd/synth.py.  The yardstick for
    the HIP path is the reference's OWN float32 uncertainty on this ill-conditioned input: the same evaluation repeated with 1 / 2 /
    4 CPU threads (only the GEMM summation order changes) and with every float input moved by one ulp (8 draws) -- ``ref_spread`` =
    the largest deviation of those runs from the stored one.  (A float64 evaluation is no anchor here: the distogram's strict bin
    tests flip between float32 and float64 distances.)  Also stored: the largest activation the reference sees (forward hooks)."""
    from str2str_amd.synth import synth_state_dict

    net, manifest = G.build_net(seed=0, sigma_final=0.02)
    net.load_state_dict(synth_state_dict(manifest, seed=0, sigma_final=0.02, style="trained_like"), strict=True)
    g = torch.Generator().manual_seed(seed)
    batch = G.make_batch(g, B, N, True)
    amax = {}

    def hook(name):
        def f(mod, inp, outp):
            amax[name] = max(amax.get(name, 0.0), float(outp.abs().max()))
        return f

    hs = [mod.register_forward_hook(hook(name)) for name, mod in net.named_modules() if isinstance(mod, torch.nn.Linear)]
    with torch.no_grad():
        out = net(batch)
    for h in hs:
        h.remove()
    r0, p0 = out["rigids"].to_tensor_7(), out["psi"]
    spread = spread_psi = 0.0
    with torch.no_grad():
        for th in (1, 2, 4):
            torch.set_num_threads(th)
            o = net(batch)
            spread = max(spread, float((o["rigids"].to_tensor_7() - r0).abs().max()))
            spread_psi = max(spread_psi, float((o["psi"] - p0).abs().max()))
        torch.set_num_threads(8)
        gj = torch.Generator().manual_seed(5)
        for _ in range(8):
            bj = dict(batch)
            for k in ("rigids_t", "sc_ca_t"):
                x = batch[k]
                up = torch.randint(0, 3, x.shape, generator=gj) - 1          # -1, 0, +1 ulp per element
                bj[k] = torch.where(up > 0, torch.nextafter(x, x + 1), torch.where(up < 0, torch.nextafter(x, x - 1), x))
            o = net(bj)
            spread = max(spread, float((o["rigids"].to_tensor_7() - r0).abs().max()))
            spread_psi = max(spread_psi, float((o["psi"] - p0).abs().max()))
    print("reference float32 spread (threads, 1-ulp input jitter) on frames:", spread, " psi:", spread_psi)
    print("largest activations:", sorted(amax.items(), key=lambda kv: -kv[1])[:5])
    G.npz(name, **{f"in_{k}": v for k, v in batch.items()}, rigids7=r0, psi=p0,
          atom37=out["atom37"][..., :5, :], ref_spread=spread, ref_psi_spread=spread_psi, hidden_amax=max(amax.values()))


if __name__ == "__main__":
    if "--cfg3" in sys.argv:
        cfg3()
    if "--cfg4" in sys.argv:
        cfg4()
    if "--cfg5" in sys.argv:
        cfg5()
    if "--trained" in sys.argv:
        trained()
    if "--trained256" in sys.argv:
        trained(1, 256, 78, "net_b1n256_trained_like.npz")
