"""Headline benchmark: sampled conformations / second, 256-residue chain, 100 denoise steps.

    python bench.py [--gpus N --steps K --warmup W] [--config cfg2|cfg3|cfg4|cfg5]   (N > 1: torch.distributed.run)

One "step" = one replica chunk sampled end to end on every GPU: forward marginal, 1 self-conditioning forward +
S x (score-network forward + fused SE(3) step), backbone projection, RCCL gather of the coordinates to rank 0 and
copy to the host.  Replicas are independent, so the work is sharded with no data-path collective except that final
gather (weak scaling: the same replicas per GPU at every N).  Prints ONE JSON line (see README "bench contract"); adds
  roofline     the dominant kernel (s2s_edge_transition, MFMA bound) timed per launch with HIP events on the launch
               stream inside the timed region (+ `ipa_kernel`: the HBM-bound IPA core the north star names)
  cpu_baseline the CPU oracle (a port of the reference path, bit-equal to it on CPU) timed on this box's host cores on
               a bounded sample of the same workload (rank 0, N=1 only)
--config selects the BASELINE.json workload (default cfg2 = configs[1], the one the metric is quoted on):
  cfg3  configs[2]: the 12 Science2011 fast folders (tests/golden/pdb), 1000 replicas each, PDB text written natively
  cfg4  configs[3]: 512-residue chain, 128 replicas per GPU (1024 over 8), 200 steps
  cfg5  configs[4]: 32 chains U[64,384] (seed 5) x 256 replicas, FLOP-weighted length-bucketed plan; at --gpus N < 8 each
        rank runs its share of the 8-rank plan (weak scaling: 1/8 of the job per GPU)
"""
import argparse
import json
import os
import platform
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FLOPS_PER_PAIR_ET = 491520          # 2*(128*384 + 384*384 + 384*128) fp32 multiply-adds x2
MFMA_FP32_PEAK = 157.3e12           # /opt/skills/guides/MI355X_MICROARCH.md (v_mfma_f32_32x32x2_f32)
MFMA_F16_PEAK = 2500e12             # dense f16 / bf16 MFMA (v_mfma_f32_32x32x16_f16), /opt/skills/guides/MI355X_MICROARCH.md
HBM_PEAK = 8e12
MFMA_F16_SUSTAINED = 1800e12        # tools/ubench/mfma_shape_power.hip: what the 1400 W cap leaves of the nominal peak, random operands, every SIMD busy
FLOPS_PER_PAIR_EE_MATRIX = 65536    # edge embedding, layers 2 and 3 (2 x 128x128 multiply-adds x2); the first layer (120 -> 128) is a table
FLOPS_PER_PAIR_EE_SURVEY = 96256    #   lookup in the kernel -- SURVEY 8(d) counts it as the reference's GEMM: 2*(120*128 + 2*128*128)
BYTES_PER_PAIR_EE = 672             # writes: z 512 B + the first IPA block's attn_bias 32 B + pair_z 128 B (inputs are O(N))
BYTES_PER_PAIR_ET = 1013            # edge row read 512 + written 512 (+ proj outputs / mask, per-node parts amortised)
METRIC = "sampled conformations/sec (whole node), 256-res chain, 100 denoise steps"


def mfma_block(kernel, alg_flops, seconds, mode, **more):
    """SURVEY 8(d) roofline block of an MFMA-bound kernel.  ``frac`` = ALGORITHMIC flops / time / dense peak of the MFMA dtype the kernel
    issues (f16 for the split-f16 arithmetic, fp32-matrix for the exact one).  The f16x3 scheme issues every product three times
    (x_h w_h + x_h w_l + x_l w_h): that figure is ``mfma_issue_frac`` -- matrix-pipe occupancy, not useful work."""
    split = 3 if mode == "f16x3" else 1
    peak = MFMA_F16_PEAK if mode == "f16x3" else MFMA_FP32_PEAK
    ach = alg_flops / seconds
    blk = {"bound": "mfma", "kernel": kernel, "achieved": ach / 1e12, "peak": peak / 1e12, "unit": "TFLOP/s", "frac": ach / peak,
           "peak_source": ("nominal dense f16 MFMA 2.5 PFLOP/s" if mode == "f16x3" else "nominal fp32-matrix MFMA 157.3 TFLOP/s")
                          + " (/opt/skills/guides/MI355X_MICROARCH.md)",
           "mfma_issue_frac": split * ach / peak, "products_issued_per_fp32_product": split,
           "vs_fp32_matrix_peak": ach / MFMA_FP32_PEAK, "algorithmic_flops": alg_flops, "seconds": seconds}
    if mode == "f16x3":
        blk["frac_of_sustained_peak"] = ach / MFMA_F16_SUSTAINED
        blk["sustained_peak_note"] = "1.8 PFLOP/s = f16 MFMA rate this chip sustains under its power cap (tools/ubench/README.md); mfma_issue_frac x 2.5/1.8 = issue rate vs that"
    blk.update(more)
    return blk


def hbm_block(kernel, alg_bytes, seconds, **more):
    blk = {"bound": "hbm", "kernel": kernel, "achieved": alg_bytes / seconds / 1e9, "peak": HBM_PEAK / 1e9, "unit": "GB/s",
           "frac": alg_bytes / seconds / HBM_PEAK, "algorithmic_bytes": alg_bytes, "seconds": seconds}
    blk.update(more)
    return blk


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return platform.processor() or "unknown"


def cpu_baseline(n_res, denoise_steps, steps_sampled=10, replicas=4):
    """Oracle on the host cores.  First a THREAD SWEEP (the oracle is eager PyTorch: its GEMMs stop scaling long before a
    2-socket box runs out of cores, and 128 threads over two NUMA nodes measured 2x slower than 8): one pass of
    (1 self-conditioning + 1 denoise) evaluations at ``replicas`` replicas per torch.set_num_threads setting, best kept.  Then the
    sample itself at the best setting: (1 self-conditioning forward + ``steps_sampled`` denoise steps) for ``replicas`` replicas of
    the same synthetic chain, extrapolated linearly to ``denoise_steps`` steps (replicas are independent; per-evaluation cost does
    not depend on t)."""
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

    def sample(n_steps):
        torch.manual_seed(42)
        t0 = time.perf_counter()
        # num_timesteps = n_steps -> exactly n_steps network evaluations (+1 self-conditioning)
        OD.forward_backward(lambda b: ON.denoising_net(sd, b), d, f, rig0, 1.0, num_timesteps=n_steps)
        return time.perf_counter() - t0

    nproc = os.cpu_count() or 1
    keep = torch.get_num_threads()
    sweep = {}
    for nt in sorted({min(nproc, x) for x in (8, 16, 32, 64, 128)}):
        torch.set_num_threads(nt)
        sweep[nt] = sample(1) / 2 / replicas          # seconds per (replica, evaluation)
    best = min(sweep, key=sweep.get)
    torch.set_num_threads(best)
    dt = sample(steps_sampled)
    torch.set_num_threads(keep)
    per_forward = dt / (steps_sampled + 1)
    conf_per_s = replicas / (per_forward * (denoise_steps + 1))
    return {"value": conf_per_s, "unit": "conformations/s", "cores": best, "kind": "port",
            "cpu_model": cpu_model(), "nproc": nproc,
            "thread_sweep_s_per_replica_evaluation": {str(k): round(v, 4) for k, v in sweep.items()},
            "sample": f"{replicas} replicas x (1 self-conditioning + {steps_sampled} denoise) network evaluations of the "
                      f"{n_res}-residue workload = {dt:.1f} s on {best} host threads (best of the sweep)"
                      + ("" if steps_sampled == denoise_steps else f", scaled linearly to {denoise_steps}+1 evaluations (SURVEY 8d asks 2 x 100 steps: bounded here to "
                         "keep the default run in minutes; the full sample, --cpu-replicas 2 --cpu-steps 100, measured 0.01287 conformations/s = the scaled "
                         "figure: profiles/r06_cpu_baseline_full_sample.json)")}


def traffic_from_profiles(pairs, mode):
    """HBM bytes per launch of the dominant kernel.  NOT measured in this run: PMC counters need rocprofv3 around the
    process (separate --pmc passes, tools/pmc_hbm_traffic.sh), so the newest committed pass (profiles/r<round><run>_pmc_hbm_traffic[_b<B>].json;
    the file states its shape) is scaled by the pairs of this launch and labelled as such.  (None, None) if there is none."""
    import glob
    import re

    def _order(path):   # newest evidence run first: round number, then the run's letter (r06e > r06 > r05d ...)
        m = re.match(r"r(\d+)([a-z]*)_pmc_hbm_traffic(_b\d+)?\.json$", os.path.basename(path))
        return (int(m.group(1)), m.group(2)) if m else (-1, "")

    key = {"f16x3": "edge_transition_f16x3"}.get(mode, "edge_transition")
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_hbm_traffic*.json")), key=_order, reverse=True):
        if _order(path)[0] < 0:
            continue
        try:
            with open(path) as f:
                d = json.load(f)
            sh = d.get("shape", {})
            return (d["kernels"][key]["bytes_per_pair_corrected"] * pairs,
                    f"profiles/{os.path.basename(path)} (PMC pass at B={sh.get('B', '?')}, N={sh.get('N', '?')}: FETCH_SIZE x 2 + WRITE_SIZE, separate passes, "
                    "scaled per pair; mean of the trunk's three launches; the counters sit on the L2's fabric side and include Infinity-Cache hits)")
        except Exception:
            continue
    return None, None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", default="cfg2", choices=["cfg2", "cfg3", "cfg4", "cfg5", "ref_default"])
    ap.add_argument("--n-res", type=int, default=None)
    ap.add_argument("--replicas", type=int, default=None, help="replicas per GPU per step")
    ap.add_argument("--denoise-steps", type=int, default=None)
    ap.add_argument("--rng", default="device", choices=["device", "host"], help="noise source (host = reference-order parity mode)")
    ap.add_argument("--cpu-steps", type=int, default=10, help="denoise steps of the CPU-oracle sample (4 replicas at the best thread count of a sweep; ~60 s of host time at N = 256)")
    ap.add_argument("--cpu-replicas", type=int, default=4, help="replicas of the CPU-oracle sample (SURVEY 8d's full sample: --cpu-replicas 2 --cpu-steps 100, ~3 min of host time)")
    ap.add_argument("--no-other-configs", action="store_true", help="default cfg2 line at 1 GPU: do not append one step each of cfg3 / cfg4 / cfg5")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-table", action="store_true", help="skip the extra (untimed) step that times the kernel families")
    a = ap.parse_args()

    import torch.distributed as dist

    from str2str_amd import ops
    from str2str_amd.common.rigid_utils import Rigid
    from str2str_amd.factory import build_diffuser, build_synthetic_net
    from str2str_amd.sampler import (forward_backward, forward_backward_chunks, forward_backward_deltas, merge_chunk_groups, merge_delta_groups,
                                     plan_mixed_work, sample_mixed_lengths, schedule)
    from str2str_amd.synth import synth_chain

    from str2str_amd.utils.launch import in_distributed_job, relaunch

    # One command, any device count (as the reference's `trainer=ddp` run, src/eval.py:129,154): from a bare shell `--gpus N` starts its
    # own N ranks under torch.distributed.run; inside a job (the driver's launch line, WORLD_SIZE set) this IS one of the ranks.
    use_dist = in_distributed_job()
    if a.gpus > 1 and not use_dist:
        if os.environ.get("S2S_BENCH_BACKEND", "nccl") == "nccl" and torch.cuda.device_count() < a.gpus:
            sys.exit(f"bench.py: --gpus {a.gpus} but {torch.cuda.device_count()} GPU(s) are visible here (RCCL wants one device per rank)")
        sys.exit(relaunch(a.gpus, __file__, sys.argv[1:]))
    world = int(os.environ.get("WORLD_SIZE", "1")) if use_dist else 1
    rank = int(os.environ.get("RANK", "0")) if use_dist else 0
    local = int(os.environ.get("LOCAL_RANK", "0")) if use_dist else 0
    if world != a.gpus:
        sys.exit(f"bench.py: --gpus {a.gpus} but this process is rank {rank} of a {world}-rank torch.distributed.run job "
                 f"(pass --gpus {world}, or start it from a bare shell and let bench.py launch its own ranks)")
    # S2S_BENCH_BACKEND=gloo is a TEST hook: it lets the multi-rank control flow (rendezvous, barriers, max-over-ranks
    # timing, gather, rank-0-only output) run with several ranks sharing one GPU; the measured configuration is nccl (RCCL).
    backend = os.environ.get("S2S_BENCH_BACKEND", "nccl")
    local = local % torch.cuda.device_count() if backend == "gloo" else local
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if use_dist:   # (also a 1-rank job: `torch.distributed.run --nproc-per-node 1` runs the RCCL path end to end on one GPU)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)  # RCCL over xGMI
        else:
            dist.init_process_group(backend)
        # the host part of a step is tiny: keep the ranks from oversubscribing the cores
        torch.set_num_threads(max(1, (os.cpu_count() or world) // world))
    ops.load_library()
    # Preflight of the collective backend: every rank contributes (rank, device UUID); what comes back proves that `world` distinct
    # processes on `world` distinct devices are in the job before anything is timed (reported as distributed.ranks_seen / devices_seen).
    uuid = str(getattr(torch.cuda.get_device_properties(dev), "uuid", f"device{local}"))
    seen = [(rank, uuid)]
    if use_dist:
        seen = [None] * world
        dist.all_gather_object(seen, (rank, uuid))

    defaults = {"cfg2": (256, 128, 100), "cfg3": (None, 1000, 100), "cfg4": (512, 128, 200), "cfg5": (None, 256, 100),
                "ref_default": (None, 100, 1000)}[a.config]
    N = a.n_res if a.n_res is not None else defaults[0]
    B = a.replicas if a.replicas is not None else defaults[1]
    S = a.denoise_steps if a.denoise_steps is not None else defaults[2]
    net = build_synthetic_net(seed=0, sigma_final=0.002, device=dev)
    diff = build_diffuser(os.path.join("/tmp", f"str2str_cache_{rank}"))
    gdev = dev if backend == "nccl" else torch.device("cpu")
    extra = {}

    # ---------------------------------------------------------------- the workload of one step on this rank
    if a.config in ("cfg2", "cfg4"):
        feats = synth_chain(N)
        rig0 = Rigid.from_tensor_4x4(feats["rigidgroups_gt_frames"][..., 0, :, :].repeat(B, 1, 1, 1))
        # the gather carries the compact backbone [B, N, 5, 3] (N, CA, C, CB, O: everything compute_backbone fills of atom37's 37
        # slots, all_atom.py:141-173): 2.0 instead of 14.5 MB per rank at cfg2
        gathered = [torch.empty(B, N, 5, 3, device=gdev) for _ in range(world)] if (use_dist and rank == 0) else None
        per_rank = B
        workload = (f"configs[{1 if a.config == 'cfg2' else 3}]: single {N}-residue synthetic chain, {B} replicas per GPU x {S} "
                    f"denoise steps (+1 self-conditioning forward), probability-flow ODE, seeded synthetic weights")
        pairs_main = B * N * N
        eval_pairs, eval_ipa_bytes = B * N * N, B * 4 * (9512 * N + 40 * N * N)   # per network evaluation, summed over the step's chunks

        def one_step(seed):
            torch.manual_seed(seed * 1000 + rank)
            torch.cuda.manual_seed(seed * 1000 + rank)  # independent noise per rank and step
            atom37 = forward_backward(net, diff, feats, rig0, 1.0, num_timesteps=S, min_t=0.01, probability_flow=True,
                                      self_conditioning=True, device=dev, rng=a.rng)
            bb = atom37[..., :5, :].contiguous()
            if use_dist:
                dist.gather(bb.to(gdev), gathered, dst=0)
                out = torch.stack(gathered) if rank == 0 else bb
            else:
                out = bb
            return out.cpu() if rank == 0 else None  # coordinates on the host of rank 0 = end of the job
    elif a.config == "cfg3":
        from str2str_amd.common import protein
        from str2str_amd.data.components.dataset import ProteinFeatureTransform

        pdb_dir = os.path.join(ROOT, "tests", "golden", "pdb")
        tf = ProteinFeatureTransform()
        targets = []
        for fn in sorted(os.listdir(pdb_dir)):
            f = tf(protein.from_pdb_string(open(os.path.join(pdb_dir, fn)).read()).to_dict())
            targets.append({k: (v[None] if torch.is_tensor(v) else v) for k, v in f.items()})
        lens = [int(t["aatype"].shape[1]) for t in targets]
        per_rank = B * len(targets)
        workload = (f"configs[2]: Science2011 fast-folder set ({len(targets)} targets, N = {sorted(lens)}), {B} replicas each per GPU "
                    f"x {S} denoise steps, one chunk per target, multi-MODEL PDB text written by the native writer")
        pairs_main = B * max(lens) ** 2
        eval_pairs, eval_ipa_bytes = sum(B * n * n for n in lens), sum(B * 4 * (9512 * n + 40 * n * n) for n in lens)
        out_dir = f"/tmp/s2s_bench_cfg3_{rank}"
        os.makedirs(out_dir, exist_ok=True)
        extra["pdb_write_s"] = 0.0

        from str2str_amd.common.pdb_utils import AsyncPdbWriter

        def one_step(seed):
            torch.manual_seed(seed * 1000 + rank)
            torch.cuda.manual_seed(seed * 1000 + rank)
            res = None
            writer = AsyncPdbWriter()   # as predict_step does: a target's file is written while the next target is sampled
            for ti, tg in enumerate(targets):
                rig0 = Rigid.from_tensor_4x4(tg["rigidgroups_gt_frames"][..., 0, :, :].repeat(B, 1, 1, 1))
                a37 = forward_backward(net, diff, tg, rig0, 1.0, num_timesteps=S, min_t=0.01, probability_flow=True,
                                       self_conditioning=True, device=dev, rng=a.rng)
                if use_dist:
                    bufs = [torch.empty_like(a37, device=gdev) for _ in range(world)] if rank == 0 else None
                    dist.gather(a37.to(gdev), bufs, dst=0)
                    a37 = torch.cat(bufs) if rank == 0 else a37
                if rank == 0:
                    res = a37
                    writer.submit(a37, os.path.join(out_dir, f"t{ti}.pdb"), aatype=tg["aatype"][0].numpy(),
                                  residue_index=tg["residue_index"][0].numpy(), chain_index=tg["chain_index"][0].numpy())
            t0 = time.perf_counter()      # what is left of the writing after the last trajectory (inside the timed region)
            writer.results()
            writer.close()
            extra["pdb_write_s"] += time.perf_counter() - t0
            return res.cpu() if res is not None else None
    elif a.config == "ref_default":
        # The reference's DEFAULT inference block (configs/model/diffusion.yaml:88-100): n_replica 100 in chunks of replica_per_batch 64
        # (64 + 36), t_delta 0.25 .. 0.70 in steps of 0.05 (10 values), num_timesteps 1000 (=> 250 .. 700 steps per t_delta), on two
        # synthetic targets of the Science2011 size range (N = 35 and 80): the regime users run, between the HIP-graph regime of tiny
        # chunks and the GPU-bound one.  B = replicas per target, S = num_timesteps.
        lens = [35, 80]
        targets = [synth_chain(n, frame_seed=3 + n, aatype_seed=4 + n) for n in lens]
        deltas = [round(0.25 + 0.05 * k, 2) for k in range(10)]
        chunks = [64] * (B // 64) + ([B % 64] if B % 64 else [])
        per_rank = B * len(targets) * len(deltas)
        n_eval = sum(int(S * d) + 1 for d in deltas)
        workload = (f"reference default inference block: {len(targets)} synthetic targets N = {lens}, {B} replicas each in chunks of {chunks}, "
                    f"t_delta {deltas[0]} .. {deltas[-1]} ({len(deltas)} values), num_timesteps {S}: {n_eval} network evaluations per chunk")
        pairs_main = 64 * max(lens) ** 2
        eval_pairs = eval_ipa_bytes = 0
        groups = {n: [[c[0] for c in g] for g in merge_chunk_groups([(c, 0, c) for c in chunks], n)] for n in lens}
        extra["evaluations_per_step"] = n_eval * sum(len(g) for g in groups.values())
        extra["trajectories_per_t_delta"] = {str(n): g for n, g in groups.items()}   # chunks sampled as one trajectory each
        extra["hip_graph"] = os.environ.get("S2S_HIP_GRAPH", "auto")
        dsteps = [schedule(d, S, 0.01)[1] for d in deltas]
        dgroups = {n: merge_delta_groups(dsteps, B, n) for n in lens}
        extra["t_deltas_per_batch"] = {str(n): [[deltas[i] for i in g] for g in dg] for n, dg in dgroups.items()}   # t_deltas sampled as one growing batch
        extra["evaluations_per_step_merged"] = sum(max(dsteps[i] for i in g) + len(g) if len(g) > 1 else (dsteps[g[0]] + 1) * len(groups[n])
                                                   for n, dg in dgroups.items() for g in dg)

        def one_step(seed):
            torch.manual_seed(seed * 1000 + rank)
            torch.cuda.manual_seed(seed * 1000 + rank)
            res = None
            for tg in targets:
                # as DiffusionLitModule.predict_step runs it: the chunks are the unit of the reference's noise stream; the chunks and the
                # t_deltas of a target whose replicas fit the pair budget are sampled as ONE growing batch (sampler.forward_backward_deltas;
                # S2S_MERGE_DELTAS=0: one t_delta at a time, S2S_MERGE_CHUNKS=0: one chunk at a time -- the reference's control flow)
                a37 = forward_backward_deltas(net, diff, tg, tg["rigidgroups_gt_frames"][..., 0, :, :], [(c, 0, c) for c in chunks], deltas,
                                              num_timesteps=S, min_t=0.01, probability_flow=True, self_conditioning=True,
                                              device=dev, rng=a.rng)
                res = a37[-1][..., :5, :]
            return res.cpu() if rank == 0 else None
    else:  # cfg5
        lens = [int(x) for x in np.random.default_rng(5).integers(64, 385, size=32)]
        targets = [synth_chain(n, frame_seed=3 + n, aatype_seed=4 + n) for n in lens]
        plan_world = max(8, world)
        plan = plan_mixed_work(lens, B, plan_world)
        my = rank % plan_world
        per_rank = sum(hi - lo for b in plan[my] for _, lo, hi in b["items"])
        workload = (f"configs[4]: 32 synthetic chains N ~ U[64,384] (seed 5) x {B} replicas x {S} denoise steps, FLOP-weighted "
                    f"length-bucketed plan over {plan_world} ranks; every GPU runs one rank's share "
                    f"({per_rank} (chain, replica) items in {len(plan[my])} padded batches)")
        pairs_main = max(sum(hi - lo for _, lo, hi in b["items"]) * b["n_pad"] ** 2 for b in plan[my])
        my_items = [(lens[k], hi - lo) for b in plan[my] for k, lo, hi in b["items"]]
        eval_pairs = sum(r * n * n for n, r in my_items)                                   # REAL pairs (padding is overhead, not work)
        eval_ipa_bytes = sum(r * 4 * (9512 * n + 40 * n * n) for n, r in my_items)
        extra["padded_pairs_over_real"] = sum(sum(hi - lo for _, lo, hi in b["items"]) * b["n_pad"] ** 2 for b in plan[my]) / max(eval_pairs, 1)

        def one_step(seed):
            torch.manual_seed(seed * 1000 + rank)
            torch.cuda.manual_seed(seed * 1000 + rank)
            pieces = sample_mixed_lengths(net, diff, targets, B, 1.0, num_timesteps=S, device=dev, rng="device",
                                          shard=(my, plan_world), plan=plan)
            # compact backbone [.,5,3] per piece, flattened: ONE gather of a padded flat buffer per step
            flat = torch.cat([p[..., :5, :].reshape(-1) for ps in pieces for _, p in ps]) if per_rank else torch.zeros(0, device=dev)
            if use_dist:
                size = torch.tensor([flat.numel()], device=gdev)
                sizes = [torch.zeros_like(size) for _ in range(world)]
                dist.all_gather(sizes, size)
                mx = int(max(int(x) for x in sizes))
                pad = torch.zeros(mx, device=gdev)
                pad[: flat.numel()] = flat.to(gdev)
                bufs = [torch.empty_like(pad) for _ in range(world)] if rank == 0 else None
                dist.gather(pad, bufs, dst=0)
                flat = torch.cat([b[: int(n)] for b, n in zip(bufs, sizes)]) if rank == 0 else flat
            return flat.cpu() if rank == 0 else None

    def barrier():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    # one-time work that belongs to no step -- packing the weights for the kernels (once per load_state_dict, ~1 s), loading the code
    # objects, the SO(3) tables -- happens here on a tiny trajectory, so that a run with --warmup 0 (the other_configs of the default
    # line) times steps and not the start of the process
    _f = synth_chain(32)
    forward_backward(net, diff, _f, Rigid.from_tensor_4x4(_f["rigidgroups_gt_frames"][..., 0, :, :].repeat(2, 1, 1, 1)), 1.0,
                     num_timesteps=2, min_t=0.01, probability_flow=True, self_conditioning=True, device=dev, rng=a.rng)
    for w in range(a.warmup):
        one_step(w)
    extra = {k: (0.0 if k == "pdb_write_s" else v) for k, v in extra.items()}
    barrier()
    t0 = time.perf_counter()
    timed = ("s2s_edge_transition", "s2s_ipa_attention") if a.config in ("cfg2", "cfg4") else ()  # (timers disable HIP graphs)
    with ops.KernelTimer(*timed) as kt:
        for k in range(a.steps):
            res = one_step(100 + k)
        my_elapsed = time.perf_counter() - t0   # this rank's own time (before the closing barrier)
        barrier()
        elapsed = time.perf_counter() - t0
    el = torch.tensor([elapsed], device=gdev, dtype=torch.float64)
    mine = torch.tensor([my_elapsed], device=gdev, dtype=torch.float64)
    per_rank_s = [mine.clone() for _ in range(world)]
    if use_dist:
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
        dist.all_gather(per_rank_s, mine)
    elapsed = float(el.item())
    et_ms, et_n = kt.mean_ms("s2s_edge_transition") if timed else (float("nan"), 0)
    ipa_ms, ipa_n = kt.mean_ms("s2s_ipa_attention") if timed else (float("nan"), 0)
    # Kernel-family time table: ONE more step, outside the timed region (per-launch HIP events on the launch stream; they switch the
    # HIP-graph replay of tiny chunks off, which is why cfg3 / cfg5 do not carry them inside the timed region)
    fam = ("s2s_edge_transition", "s2s_edge_embed", "s2s_ipa_attention", "s2s_encoder_attention", "s2s_node_linear")
    table = None
    if not a.no_kernel_table:   # (every rank runs the step: it contains the job's collectives; rank 0's table is reported)
        torch.cuda.synchronize()
        tp0 = time.perf_counter()
        with ops.KernelTimer(*fam) as kp:
            one_step(999)
            torch.cuda.synchronize()
            prof_s = time.perf_counter() - tp0
            table = {n: dict(zip(("total_ms", "launches"), kp.total_ms(n))) for n in fam}
        table["step_wall_ms_with_events"] = 1e3 * prof_s

    if rank == 0:
        assert res is not None and torch.isfinite(res).all()
        total = a.steps * per_rank * world
        from str2str_amd.arith import net_arith

        mode = net_arith(net)
        line = {
            "metric": METRIC if a.config == "cfg2" else f"sampled conformations/sec (whole node), BASELINE {a.config}",
            "value": total / elapsed, "unit": "conformations/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": 1e3 * elapsed / a.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": {"f32": "f32",
                      "f16x3": "f32 (matrix products on 2-way f16 split MFMA 'f16x3': 3 products per block, fp32 accumulate, fp32-equivalent)"}[mode],
            "data": "synthetic",
            "config": {"workload": workload, "n_res": N, "replicas_per_gpu": B, "denoise_steps": S,
                       "parallelism": f"replica-shard x{world}", "arith": mode, "range_fallback": sorted(getattr(net, "range_fallback", None) or ()),
                       "range_headroom": ops.range_headroom(), "rng": a.rng,
                       "rng_note": "device Philox noise (throughput mode): checked for finite results and the forward-marginal distribution; "
                                   "the fixed-seed parity runs of tests/ use --rng host" if a.rng == "device" else "host generator, reference draw order",
                       "step_definition": "one replica chunk (cfg3: all 12 targets; cfg5: the rank's plan) sampled end to end incl. gather + D2H"},
            "distributed": {"backend": (dist.get_backend() if use_dist else None), "world_size": world,
                            "ranks_seen": sorted(int(r) for r, _ in seen), "devices_seen": len({u for _, u in seen}),
                            "per_rank_conformations_per_s": [a.steps * per_rank / float(x.item()) for x in per_rank_s]},
        }
        line["config"].update(extra)
        ipa_name = ("s2s_ipa_attention_f16w" if mode == "f16x3" else "s2s_ipa_attention") + " + s2s_ipa_opair"
        folded = mode == "f16x3" and os.environ.get("S2S_IPA_FOLD", "1") != "0"
        et_name = "s2s_edge_transition" + {"f16x3": "_f16x3 (edge_transition_f16_kernel)"}.get(mode, " (edge_transition_kernel)")
        if a.config in ("cfg2", "cfg4") and et_n:
            # the dominant kernel, per launch, from HIP events on the launch stream INSIDE the timed region
            pairs = pairs_main
            traffic, traffic_src = traffic_from_profiles(pairs, mode)
            line["roofline"] = mfma_block(
                et_name, pairs * FLOPS_PER_PAIR_ET, et_ms * 1e-3, mode,
                # `traffic` (HBM bytes per launch from PMC counters) cannot be collected inside this process: the counters need rocprofv3
                # around it, in passes of their own (tools/pmc_hbm_traffic.sh).  The committed pass of the same kernel is quoted
                # under its own name, scaled per pair.
                traffic=None, traffic_from_profile=traffic, traffic_source=traffic_src, launches_timed=et_n, mean_launch_ms=et_ms,
                pairs_per_launch=pairs, flops_per_pair=FLOPS_PER_PAIR_ET, algorithmic_bytes_per_launch=pairs * BYTES_PER_PAIR_ET,
                formula="frac = pairs_per_launch x flops_per_pair / mean_launch_ms / peak (SURVEY 8d; 491 520 = 2 x (128x384 + 384x384 + 384x128) x 2 "
                        "minus nothing: the G_j pre-product is O(N))")
            ipa_bytes = B * 4 * (9512 * N + 40 * N * N)
            ipa_moved = B * 4 * (5928 * N + 40 * N * N) if folded else ipa_bytes
            line["ipa_kernel"] = hbm_block(
                ipa_name, ipa_bytes, ipa_ms * 1e-3, mean_launch_ms=ipa_ms, launches_timed=ipa_n, algorithmic_bytes_per_launch=ipa_bytes,
                # SURVEY 8(d) counts the operator as the reference states it (q, k, v per head).  The default f16 path folds W_k / W_v
                # away and reads s as K and V of every head (models/net/ipa.py _folded_packs): what the launch must move then is 3584
                # floats per residue less -- `frac_moved_bytes` prices the kernel on those
                operands="K = V = s (folded projections)" if folded else "per-head k / v",
                moved_bytes_per_launch=ipa_moved, frac_moved_bytes=ipa_moved / (ipa_ms * 1e-3) / HBM_PEAK,
                formula="frac = B x 4 x (9512 N + 40 N^2) / mean_launch_ms / 8 TB/s (SURVEY 8d)")
        if table is not None:
            line["kernel_times"] = table
            n_eval = S + 1
            et, ipa, ee, nl = (table[k] for k in ("s2s_edge_transition", "s2s_ipa_attention", "s2s_edge_embed", "s2s_node_linear"))
            src = "kernel_times (one extra step with per-launch HIP events, HIP graphs off)"
            if a.config in ("cfg3", "cfg5") and et["launches"] and ipa["launches"]:
                # efficiency statements for the small-protein / mixed-length regimes, on the REAL (unpadded) work of the step:
                # 3 edge transitions and 4 attention launches per network evaluation
                line["roofline"] = mfma_block(et_name, 3 * n_eval * eval_pairs * FLOPS_PER_PAIR_ET, et["total_ms"] * 1e-3, mode, traffic=None,
                                              launches_timed=et["launches"], total_ms=et["total_ms"], source=src)
                ib = 4 * n_eval * eval_ipa_bytes
                line["ipa_kernel"] = hbm_block(ipa_name, ib, ipa["total_ms"] * 1e-3, launches_timed=ipa["launches"], total_ms=ipa["total_ms"], source=src)
            if eval_pairs and ee["launches"]:
                # edge embedding: one launch per evaluation; both of its roofs (it is bound by neither: VALU issue)
                eef = n_eval * eval_pairs * FLOPS_PER_PAIR_EE_MATRIX
                blk = mfma_block("s2s_edge_embed" + ("_f16x3 (edge_embed_f16_kernel)" if mode == "f16x3" else ""), eef, ee["total_ms"] * 1e-3, mode,
                                 launches_timed=ee["launches"], total_ms=ee["total_ms"], flops_per_pair=FLOPS_PER_PAIR_EE_MATRIX, source=src,
                                 flops_per_pair_survey_8d=FLOPS_PER_PAIR_EE_SURVEY,
                                 note="the 120 -> 128 first layer is a table lookup in the kernel: matrix work = layers 2 and 3")
                blk["frac_survey_flops"] = blk["frac"] * FLOPS_PER_PAIR_EE_SURVEY / FLOPS_PER_PAIR_EE_MATRIX
                blk["hbm"] = hbm_block("(same launches)", n_eval * eval_pairs * BYTES_PER_PAIR_EE, ee["total_ms"] * 1e-3, bytes_per_pair=BYTES_PER_PAIR_EE)
                line["edge_embed_kernel"] = blk
            if nl["launches"] and kp.work.get("s2s_node_linear", [0])[0]:
                line["node_kernel"] = mfma_block("s2s_node_linear / _multi / _chain / _vfrag (node_gemm.hip)", kp.work["s2s_node_linear"][0],
                                                 nl["total_ms"] * 1e-3, mode, launches_timed=nl["launches"], total_ms=nl["total_ms"], source=src,
                                                 note="flops = sum over launches of 2 M K N as launched; ~11 launches of 0.1 ms each per IPA block at cfg2: "
                                                      "launch- and latency-bound, the fraction is reported for completeness")
        if world == 1 and not a.no_cpu_baseline and a.config == "cfg2":
            line["cpu_baseline"] = cpu_baseline(N, S, steps_sampled=a.cpu_steps, replicas=a.cpu_replicas)
        if world == 1 and a.config == "cfg2" and not a.no_other_configs and a.n_res is None and a.replicas is None and a.denoise_steps is None:
            # the other single-GPU workloads of BASELINE.json and the reference's default inference block (configs/model/diffusion.yaml:88-100
            # there: the workload its users run), ONE step each (their own processes, after the timed region and the CPU baseline):
            # driver-visible numbers for cfg3 / cfg4 / cfg5 / ref_default beside the headline
            import subprocess

            torch.cuda.empty_cache()
            line["other_configs"] = {}
            for cfg in ("cfg3", "cfg4", "cfg5", "ref_default"):
                t_sub = time.perf_counter()
                try:
                    # (cfg4 times its kernels inside the timed region like cfg2; cfg3 / cfg5 run under HIP graphs, so their kernel
                    #  fractions come from one more step with per-launch events: the kernel table)
                    r = subprocess.run([sys.executable, os.path.abspath(__file__), "--config", cfg, "--steps", "1", "--warmup", "0",
                                        "--no-cpu-baseline"] + ([] if cfg in ("cfg3", "cfg5") else ["--no-kernel-table"]), capture_output=True, text=True, timeout=900, cwd=ROOT,
                                       env={k: v for k, v in os.environ.items() if k not in   # (side runs are plain 1-GPU processes)
                                            ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "GROUP_RANK", "LOCAL_WORLD_SIZE", "TORCHELASTIC_RUN_ID")})
                    sub = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
                    line["other_configs"][cfg] = {"value": sub["value"], "unit": sub["unit"], "ms_per_step": sub["ms_per_step"], "steps": 1,
                                                  "warmup": 0, "workload": sub["config"]["workload"],
                                                  "range_fallback": sub["config"].get("range_fallback"),
                                                  **{k: {kk: vv for kk, vv in sub[k].items() if kk in ("kernel", "bound", "achieved", "peak", "unit", "frac", "mfma_issue_frac",
                                                                                                      "frac_moved_bytes", "launches_timed", "mean_launch_ms", "total_ms")}
                                                     for k in ("roofline", "ipa_kernel", "edge_embed_kernel") if k in sub},
                                                  "process_wall_s": round(time.perf_counter() - t_sub, 1)}
                except Exception as e:   # a failed side run must not cost the headline line
                    line["other_configs"][cfg] = {"error": f"{type(e).__name__}: {e}"[:300]}
        print(json.dumps(line), flush=True)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
