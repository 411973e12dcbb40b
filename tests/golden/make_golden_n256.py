"""One network evaluation of the REFERENCE at the bench shape (BASELINE configs[1]: N = 256), B = 1, with the same
seeded synthetic weights as make_golden.py -> tests/golden/net_b1n256.npz (inputs + frames / psi / backbone).
Run in the authoring container (needs /root/reference):  python tests/golden/make_golden_n256.py  (~1 min on CPU)."""
import torch

import make_golden as G  # noqa: E402  (sets up the reference import shim)


def main():
    net, _ = G.build_net(seed=0, sigma_final=0.02)
    g = torch.Generator().manual_seed(140)
    batch = G.make_batch(g, 1, 256, False)
    with torch.no_grad():
        out = net(batch)
    G.npz("net_b1n256.npz", **{f"in_{k}": v for k, v in batch.items()}, rigids7=out["rigids"].to_tensor_7(), psi=out["psi"],
          atom37=out["atom37"][..., :5, :], atom14=out["atom14"][..., :5, :])


def trajectory():
    """Free-running reference trajectory at N = 256 (B = 2 -> 1024 tiles of 128 pairs: every persistent workgroup of the pair
    kernels walks several tiles), 5 denoise steps, contractive weights -> tests/golden/traj_free_n256_s5.npz (~2 min)."""
    from str2str_amd.synth import synth_chain

    diff = G.build_diffuser()
    net2, _ = G.build_net(seed=0, sigma_final=0.002)
    N, B, S, td = 256, 2, 5, 1.0
    feats = synth_chain(N)
    rig0 = G.Rigid.from_tensor_4x4(feats["rigidgroups_gt_frames"][..., 0, :, :].clone().repeat(B, 1, 1, 1))
    torch.manual_seed(42)
    trace = []
    atom37, ts, dt = G.ref_forward_backward(net2, diff, feats, rig0, td, num_timesteps=S, trace=trace)
    G.npz("traj_free_n256_s5.npz", atom37=atom37[..., :5, :], ts=ts.copy(), dt=dt, seed=42, n_res=N, B=B, num_timesteps=S,
          t_delta=td, first_rigids_t=trace[0]["rigids_t"], last_x0=trace[-1]["x0"])


def trajectory100():
    """The HEADLINE workload as the reference runs it: free-running forward-backward trajectory at N = 256, 100 denoise steps
    (+1 self-conditioning evaluation), B = 2 replicas, t_delta = 1.0, probability-flow ODE, contractive weights (sigma_final
    = 0.002) -> tests/golden/traj_free_n256_s100.npz: final backbone coordinates + the frames entering steps 25 / 50 / 75 /
    the last one (so a divergence can be located).  ~6 min on 8 CPU threads."""
    from str2str_amd.synth import synth_chain

    diff = G.build_diffuser()
    net2, _ = G.build_net(seed=0, sigma_final=0.002)
    N, B, S, td = 256, 2, 100, 1.0
    feats = synth_chain(N)
    rig0 = G.Rigid.from_tensor_4x4(feats["rigidgroups_gt_frames"][..., 0, :, :].clone().repeat(B, 1, 1, 1))
    torch.manual_seed(42)
    trace = []
    atom37, ts, dt = G.ref_forward_backward(net2, diff, feats, rig0, td, num_timesteps=S, trace=trace)
    G.npz("traj_free_n256_s100.npz", atom37=atom37[..., :5, :], ts=ts.copy(), dt=dt, seed=42, n_res=N, B=B, num_timesteps=S,
          t_delta=td, first_rigids_t=trace[0]["rigids_t"], last_x0=trace[-1]["x0"],
          **{f"rigids_t_step{k}": trace[k]["rigids_t"] for k in (25, 50, 75, 99)})


if __name__ == "__main__":
    import sys

    if "--trajectory100" in sys.argv:
        trajectory100()
    elif "--trajectory" in sys.argv:
        trajectory()
    else:
        main()
