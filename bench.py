"""Headline benchmark: sampled conformations / second, 256-residue chain, 100 denoise steps.

    python bench.py [--gpus N --steps K --warmup W]          (N > 1: launched by torch.distributed.run)

One "step" = one replica chunk sampled end to end on every GPU: forward marginal (host noise),
1 self-conditioning forward + 100 x (score-network forward + fused SE(3) step), backbone projection,
RCCL gather of the coordinates to rank 0 and copy to the host.  Replicas are independent, so the
work is sharded with no data-path collective except that final gather (weak scaling: 128 replicas
per GPU).  Prints ONE JSON line (see README "bench contract"); adds
  roofline     the dominant kernel (s2s_edge_transition, fp32 MFMA bound) timed per launch with
               HIP events on the launch stream inside the timed region
  cpu_baseline the CPU oracle (a port of the reference path, bit-equal to it on CPU) timed on this
               box's host cores on a bounded sample of the same workload (rank 0, N=1 only)
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

N_RES, REPLICAS, DENOISE_STEPS = 256, 128, 100
FLOPS_PER_PAIR_ET = 491520          # 2*(128*384 + 384*384 + 384*128) fp32 multiply-adds x2
MFMA_FP32_PEAK = 157.3e12           # /opt/skills/guides/MI355X_MICROARCH.md (v_mfma_f32_32x32x2_f32)
MFMA_BF16_PEAK = 2500e12            # dense bf16 MFMA (v_mfma_f32_32x32x16_bf16)


def cpu_baseline(n_res, steps_sampled=5, replicas=2):
    """Oracle on the host cores: (1 self-conditioning forward + `steps_sampled` denoise steps) for
    `replicas` replicas of the same synthetic chain, extrapolated linearly to DENOISE_STEPS steps."""
    from oracle import diffuser as OD
    from oracle import geometry as OG
    from oracle import net as ON
    from str2str_amd.synth import synth_chain, synth_state_dict
    from str2str_amd.factory import build_net

    feats = synth_chain(n_res)
    manifest = [(k, tuple(v.shape)) for k, v in build_net().state_dict().items()]
    sd = synth_state_dict(manifest, seed=0, sigma_final=0.002)
    f = {k: v.repeat(replicas, *(1,) * (v.ndim - 1)) for k, v in feats.items()
         if k in ("aatype", "residue_mask", "fixed_mask", "residue_idx", "torsion_angles_sin_cos")}
    rig0 = OG.Frames.from_tensor_4x4(feats["rigidgroups_gt_frames"][..., 0, :, :].repeat(replicas, 1, 1, 1))
    d = OD.FrameDiffuser()
    torch.manual_seed(42)
    t0 = time.perf_counter()
    # num_timesteps = steps_sampled -> exactly steps_sampled network evaluations (+1 self-conditioning)
    OD.forward_backward(lambda b: ON.denoising_net(sd, b), d, f, rig0, 1.0, num_timesteps=steps_sampled)
    dt = time.perf_counter() - t0
    per_forward = dt / (steps_sampled + 1)
    conf_per_s = replicas / (per_forward * (DENOISE_STEPS + 1))
    return {"value": conf_per_s, "unit": "conformations/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"{replicas} replica x (1 self-conditioning + {steps_sampled} denoise) network evaluations of the "
                      f"{n_res}-residue workload = {dt:.1f} s on the host, scaled linearly to {DENOISE_STEPS}+1 evaluations"}


def traffic_bytes(pairs, mode):
    """HBM bytes per launch of the dominant kernel from the committed PMC passes (profiles/*_pmc_hbm_traffic.json:
    rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate runs, read side doubled per the gfx950 correction; recipe
    tools/pmc_hbm_traffic.sh), scaled by the pairs of this launch.  None if the file is absent."""
    name, key = ("r01i_pmc_hbm_traffic.json", "edge_transition_bf16x6") if mode == "bf16x6" else ("r01_pmc_hbm_traffic.json", "edge_transition")
    try:
        with open(os.path.join(ROOT, "profiles", name)) as f:
            return json.load(f)["kernels"][key]["bytes_per_pair_corrected"] * pairs
    except Exception:
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--n-res", type=int, default=N_RES)
    ap.add_argument("--replicas", type=int, default=REPLICAS, help="replicas per GPU per step")
    ap.add_argument("--denoise-steps", type=int, default=DENOISE_STEPS)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    a = ap.parse_args()

    import torch.distributed as dist

    from str2str_amd import ops
    from str2str_amd.common.rigid_utils import Rigid
    from str2str_amd.factory import build_diffuser, build_synthetic_net
    from str2str_amd.sampler import forward_backward
    from str2str_amd.synth import synth_chain

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == a.gpus, f"--gpus {a.gpus} but WORLD_SIZE={world} (launch with torch.distributed.run)"
    # S2S_BENCH_BACKEND=gloo is a TEST hook: it lets the multi-rank control flow (rendezvous, barriers, max-over-ranks
    # timing, gather, rank-0-only output) run with several ranks sharing one GPU; the measured configuration is nccl (RCCL).
    backend = os.environ.get("S2S_BENCH_BACKEND", "nccl")
    local = local % torch.cuda.device_count() if backend == "gloo" else local
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)  # RCCL over xGMI
        else:
            dist.init_process_group(backend)
        # the host part of a step (forward marginal on the host generator) is tiny: keep the ranks from oversubscribing
        torch.set_num_threads(max(1, (os.cpu_count() or world) // world))
    ops.load_library()

    N, B, S = a.n_res, a.replicas, a.denoise_steps
    feats = synth_chain(N)
    net = build_synthetic_net(seed=0, sigma_final=0.002, device=dev)
    diff = build_diffuser(os.path.join("/tmp", f"str2str_cache_{rank}"))
    rig0 = Rigid.from_tensor_4x4(feats["rigidgroups_gt_frames"][..., 0, :, :].repeat(B, 1, 1, 1))
    gdev = dev if backend == "nccl" else torch.device("cpu")
    gathered = [torch.empty(B, N, 37, 3, device=gdev) for _ in range(world)] if (world > 1 and rank == 0) else None

    def one_step(seed):
        torch.manual_seed(seed * 1000 + rank)  # independent noise per rank and step
        atom37 = forward_backward(net, diff, feats, rig0, 1.0, num_timesteps=S, min_t=0.01, probability_flow=True,
                                  self_conditioning=True, device=dev, rng="device")
        if world > 1:
            dist.gather(atom37.to(gdev), gathered, dst=0)
            out = torch.stack(gathered) if rank == 0 else atom37
        else:
            out = atom37
        return out.cpu() if rank == 0 else None  # coordinates on the host of rank 0 = end of the job

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for w in range(a.warmup):
        one_step(w)
    barrier()
    t0 = time.perf_counter()
    with ops.KernelTimer("s2s_edge_transition", "s2s_ipa_attention") as kt:
        for k in range(a.steps):
            res = one_step(100 + k)
        barrier()
        elapsed = time.perf_counter() - t0
    el = torch.tensor([elapsed], device=gdev, dtype=torch.float64)
    if world > 1:
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
    elapsed = float(el.item())
    et_ms, et_n = kt.mean_ms("s2s_edge_transition")
    ipa_ms, ipa_n = kt.mean_ms("s2s_ipa_attention")

    if rank == 0:
        assert res is not None and torch.isfinite(res).all()
        total = a.steps * B * world
        pairs = B * N * N
        mode = net.translator.trunk["edge_transition_0"].mfma_mode
        alg = pairs * FLOPS_PER_PAIR_ET                       # fp32 multiply-add flops the operator needs
        executed = alg * (6 if mode == "bf16x6" else 1)       # bf16x6: six bf16 plane-pair products per fp32 product
        peak = MFMA_BF16_PEAK if mode == "bf16x6" else MFMA_FP32_PEAK
        ach = executed / (et_ms * 1e-3)
        ipa_bytes = B * 4 * (9512 * N + 40 * N * N)
        line = {
            "metric": "sampled conformations/sec (whole node), 256-res chain, 100 denoise steps",
            "value": total / elapsed, "unit": "conformations/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": 1e3 * elapsed / a.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32" if mode == "f32" else "f32 (pair MLP on exact 3-way bf16 split MFMA, fp32 accumulate)",
            "data": "synthetic",
            "config": {"workload": f"configs[1]: single {N}-residue synthetic chain, {B} replicas per GPU x {S} denoise "
                                   f"steps (+1 self-conditioning forward), probability-flow ODE, seeded synthetic weights",
                       "n_res": N, "replicas_per_gpu": B, "denoise_steps": S, "parallelism": f"replica-shard x{world}",
                       "edge_mfma_mode": mode,
                       "step_definition": "one replica chunk sampled end to end incl. gather + D2H"},
            "roofline": {"bound": "mfma",
                         "kernel": "s2s_edge_transition" + ("_bf16x6 (edge_transition_bf16_kernel)" if mode == "bf16x6"
                                                            else " (edge_transition_kernel)"),
                         "achieved": ach / 1e12, "peak": peak / 1e12, "unit": "TFLOP/s", "frac": ach / peak,
                         "traffic": traffic_bytes(pairs, mode), "launches_timed": et_n, "mean_launch_ms": et_ms,
                         "algorithmic_flops_per_launch": alg, "executed_mfma_flops_per_launch": executed,
                         # what a bare MFMA loop on random operands sustains under this part's 1400 W cap
                         # (tools/ubench/mfma_shape_power.hip): context for `frac`, which uses the nominal peak
                         "power_limited_mfma_rate_measured": 1800.0 if mode == "bf16x6" else None,
                         "fp32_equivalent_tflops": alg / (et_ms * 1e-3) / 1e12,
                         "fp32_equivalent_vs_fp32_mfma_peak": alg / (et_ms * 1e-3) / MFMA_FP32_PEAK},
            "ipa_kernel": {"bound": "hbm", "kernel": "s2s_ipa_attention", "mean_launch_ms": ipa_ms, "launches_timed": ipa_n,
                           "achieved": ipa_bytes / (ipa_ms * 1e-3) / 1e9, "peak": 8000.0, "unit": "GB/s",
                           "frac": ipa_bytes / (ipa_ms * 1e-3) / 8e12},
        }
        if world == 1 and not a.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(N)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
