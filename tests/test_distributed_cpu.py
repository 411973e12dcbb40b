"""The N>1 path on CPU: world_size-2 (and 3) gloo processes exercise what DiffusionLitModule.predict_step / bench.py do on
RCCL: the WHOLE replica range of a (target, t_delta) sharded over the ranks, the reference's chunks walked per rank with
the host generator in lock-step (also through chunks a rank does not sample), ONE gather per t_delta, rank-major order ==
single-process MODEL order.  The network itself needs the GPU, so ``forward_backward`` is replaced by a stand-in with the
REAL function's generator behaviour (tests/test_hip_parity.py::test_sharded_* covers the real sampler on the GPU)."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_rank_chunk_slices_partition_the_replica_range():
    from str2str_amd.sampler import rank_chunk_slices

    for n, rpb, world in [(1000, 64, 8), (1024, 128, 8), (64, 64, 8), (5, 2, 3), (3, 64, 8), (1, 1, 2), (130, 64, 4), (9, 4, 8)]:
        seen, launches = [], []
        for r in range(world):
            c0 = 0
            for bsz, lo, hi in rank_chunk_slices(n, rpb, r, world):
                assert 0 <= lo <= hi <= bsz <= rpb
                seen += list(range(c0 + lo, c0 + hi))
                if hi > lo:
                    launches.append(hi - lo)
                c0 += bsz
            assert c0 == n
        assert seen == list(range(n)), (n, rpb, world)          # every replica exactly once, rank-major == replica order
    # full-size launches once every rank owns at least one chunk (cfg4: 1024 replicas, 8 ranks, chunks of 128)
    assert [hi - lo for _, lo, hi in rank_chunk_slices(1024, 128, 3, 8) if hi > lo] == [128]
    assert sorted(hi - lo for _, lo, hi in rank_chunk_slices(1000, 64, 0, 8) if hi > lo) == [61, 64]


def _fake_start_frames(diffuser, batch, rigids_0, t_delta, lo, hi, rng, device):
    """Generator behaviour of sampler._start_frames in rng='host' mode: the WHOLE chunk's forward-marginal noise is consumed whatever
    the slice; the "frames" of the slice are that noise (so a wrong slice / a generator out of lock-step shows up in the file)."""
    B, N = rigids_0.shape
    z = torch.randn(B, N, 3)              # stands for the chunk's forward-marginal draws
    return z[lo:hi] if hi > lo else None


def _fake_denoise_loop(net, diffuser, feats, rigids_t, ts, dt, *, host_noise=None, **kw):
    """A trajectory that returns its start: per step the host draws of the chunk are consumed when the caller hands them in."""
    for _ in range(len(ts) - 1):
        if host_noise is not None:
            host_noise()
    out = torch.zeros(rigids_t.shape[0], rigids_t.shape[1], 37, 3)
    out[:, :, 1, :] = rigids_t
    return out, None, None


def _fake_pass_deltas(net, diffuser, batch, groups, *, host_noise=None, **kw):
    """sampler._denoise_pass_deltas with trajectories that return their start: one host_noise() call per global step."""
    for _ in range(max(len(g["ts"]) for g in groups) - 1):
        if host_noise is not None:
            host_noise()
    outs = []
    for g in groups:
        out = torch.zeros(g["rigids_t"].shape[0], g["rigids_t"].shape[1], 37, 3)
        out[:, :, 1, :] = g["rigids_t"]
        outs.append(out)
    return outs, torch.zeros(1), None


class _Net(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.w = torch.nn.Parameter(torch.zeros(1))
        self.embedder = type("E", (), {"self_conditioning": True, "time_images": None})()


def _predict(rank, world, port, n_replica, rpb, out_dir, q):
    if world > 1:
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
        dist.init_process_group("gloo", rank=rank, world_size=world)
    os.environ["S2S_PDB_WRITER"] = "python"
    from str2str_amd.models import diffusion_module as DM
    from str2str_amd.synth import synth_chain

    from str2str_amd import sampler as SM

    # the sampler's control flow (chunks -> groups -> trajectories, slices, host-generator lock-step) runs as shipped; only the three
    # functions that need the device are replaced
    SM._start_frames, SM.denoise_loop = _fake_start_frames, _fake_denoise_loop
    SM._denoise_pass_deltas = _fake_pass_deltas                       # (the t_deltas of a target as one batch: forward_backward_deltas)
    SM._range_guarded = lambda net, run_pass, device, **kw: run_pass()
    SM._require_hip_device = lambda device, net: torch.device("cpu")
    inf = dict(n_replica=n_replica, replica_per_batch=rpb, delta_min=0.5, delta_max=0.6, delta_step=0.1, num_timesteps=4,
               noise_scale=1.0, probability_flow=True, self_conditioning=True, min_t=0.01, output_dir=out_dir, backward_only=False)
    model = DM.DiffusionLitModule(net=_Net(), diffuser=None, inference=inf)
    batch = synth_chain(6)
    batch["residue_index"] = batch["residue_index"] + 1
    torch.manual_seed(3)
    model.predict_step(batch, 0)
    if rank == 0:
        q.put("done")
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def _run_predict(world, n_replica, rpb, out_dir):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_predict, args=(r, world, port, n_replica, rpb, out_dir, q)) for r in range(world)]
    for p in procs:
        p.start()
    assert q.get(timeout=180) == "done"
    for p in procs:
        p.join(timeout=180)
        assert p.exitcode == 0
    return {d: open(os.path.join(out_dir, d, "synth6.pdb")).read() for d in ("0.5", "0.6", "all_delta")}


def test_predict_step_multi_rank_files_equal_single_process(tmp_path):
    """Whole-n_replica sharding, uneven splits (5 replicas in chunks of 2 over 2 and 3 ranks) and more ranks than replicas
    (1 replica, 2 ranks: one rank samples nothing in any chunk and must still stay in generator lock-step across the two
    t_deltas): the files rank 0 writes are byte-identical to a single-process run with the same seed."""
    for n_replica, rpb in [(5, 2), (1, 1)]:
        single = _run_predict(1, n_replica, rpb, str(tmp_path / f"w1_{n_replica}"))
        assert single["0.5"].count("MODEL ") == n_replica and single["all_delta"].count("MODEL ") == 2 * n_replica
        for world in (2, 3):
            multi = _run_predict(world, n_replica, rpb, str(tmp_path / f"w{world}_{n_replica}"))
            assert multi == single, (n_replica, rpb, world)
        # the chunks of a (target, t_delta) sampled as one trajectory (the default) against one trajectory per chunk
        os.environ["S2S_MERGE_CHUNKS"] = "0"   # (inherited by the spawned ranks)
        try:
            assert _run_predict(1, n_replica, rpb, str(tmp_path / f"w1_{n_replica}_chunkwise")) == single
            assert _run_predict(2, n_replica, rpb, str(tmp_path / f"w2_{n_replica}_chunkwise")) == single
        finally:
            del os.environ["S2S_MERGE_CHUNKS"]
        # ... and one t_delta at a time (the reference's outer loop) against the t_deltas of the target as one batch (the default)
        os.environ["S2S_MERGE_DELTAS"] = "0"
        try:
            assert _run_predict(1, n_replica, rpb, str(tmp_path / f"w1_{n_replica}_deltawise")) == single
            assert _run_predict(3, n_replica, rpb, str(tmp_path / f"w3_{n_replica}_deltawise")) == single
        finally:
            del os.environ["S2S_MERGE_DELTAS"]


def _gather_worker(rank, world, port, total, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from str2str_amd.models.diffusion_module import gather_replicas
    from str2str_amd.sampler import shard_range

    lo, hi = shard_range(total, rank, world)
    mine = torch.arange(lo, hi, dtype=torch.float32)[:, None, None, None].expand(hi - lo, 4, 37, 3).contiguous()
    out = gather_replicas(mine, total)
    if rank == 0:
        q.put(out[:, 0, 0, 0].tolist())
    else:
        assert out is None
    dist.barrier()
    dist.destroy_process_group()


def _run_gather(total, world=2):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_gather_worker, args=(r, world, port, total, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = q.get(timeout=120)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    return got


def test_gather_keeps_replica_order():
    assert _run_gather(6) == [0.0, 1.0, 2.0, 3.0, 4.0, 5.0]
    assert _run_gather(5) == [0.0, 1.0, 2.0, 3.0, 4.0]           # uneven split
    assert _run_gather(2, world=3) == [0.0, 1.0]                 # a rank with an empty slice


def _compact_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from str2str_amd.models.diffusion_module import compact_backbone, expand_backbone, gather_replicas
    from str2str_amd.sampler import shard_range

    total, N = 5, 7
    full = torch.zeros(total, N, 37, 3)
    full[:, :, :5] = torch.randn(total, N, 5, 3, generator=torch.Generator().manual_seed(11))   # what compute_backbone fills: N, CA, C, CB, O
    assert torch.equal(expand_backbone(compact_backbone(full)), full) and compact_backbone(full).shape == (total, N, 5, 3)
    lo, hi = shard_range(total, rank, world)
    out = gather_replicas(full[lo:hi], total)
    if rank == 0:
        q.put(bool(torch.equal(out, full)) and tuple(out.shape) == (total, N, 37, 3))
    dist.barrier()
    dist.destroy_process_group()


def test_gather_payload_is_the_compact_backbone_and_atom37_comes_back_exactly():
    """gather_replicas moves [., N, 5, 3] (the five slots compute_backbone fills, reference all_atom.py:141-173) and rank 0 gets the
    atom37 tensor back bit for bit -- at world 2 and in a ONE-rank group (the collective path is the same code at every world size:
    what the 1-rank RCCL test on the GPU box runs with device tensors)."""
    for world in (1, 2):
        ctx = mp.get_context("spawn")
        q = ctx.Queue()
        port = _free_port()
        procs = [ctx.Process(target=_compact_worker, args=(r, world, port, q)) for r in range(world)]
        for p in procs:
            p.start()
        assert q.get(timeout=120) is True
        for p in procs:
            p.join(timeout=120)
            assert p.exitcode == 0


def test_entry_points_start_their_own_ranks(tmp_path):
    """`bench.py --gpus N` / `eval.py trainer.devices=N` from a bare shell re-execute under torch.distributed.run (reference: Lightning
    spawns the ranks behind trainer.predict, src/eval.py:129,154): utils.launch.relaunch runs a script as N ranks on 127.0.0.1 and
    hands its exit code on; inside a job nothing is re-launched, and a --gpus / WORLD_SIZE mismatch is a clear error, not an assert."""
    import subprocess
    import sys

    from str2str_amd.utils import launch

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "who.py"
    script.write_text("import os, sys\nprint('rank', os.environ['RANK'], 'of', os.environ['WORLD_SIZE'], 'addr', os.environ['MASTER_ADDR'], sys.argv[1:], flush=True)\n"
                      "sys.exit(3 if '--fail' in sys.argv and os.environ['RANK'] == '1' else 0)\n")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    code = "import sys; from str2str_amd.utils.launch import relaunch; sys.exit(relaunch(2, sys.argv[1], sys.argv[2:]))"
    r = subprocess.run([sys.executable, "-c", code, str(script), "--x", "1"], cwd=root, env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr[-1500:]
    assert "rank 0 of 2 addr 127.0.0.1 ['--x', '1']" in r.stdout and "rank 1 of 2 addr 127.0.0.1 ['--x', '1']" in r.stdout
    r = subprocess.run([sys.executable, "-c", code, str(script), "--fail"], cwd=root, env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode != 0                                              # a failing rank fails the job
    cmd = launch.launch_command(8, "bench.py", ["--gpus", "8"], port=1234)
    assert cmd[1:8] == ["-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=8", "--master-addr", "127.0.0.1", "--master-port"]
    assert [launch.resolve_devices(d) for d in (1, "2", [0, 1, 2], "0,1", None, 4)] == [1, 2, 3, 2, 1, 4]
    assert not launch.in_distributed_job() or "WORLD_SIZE" in os.environ
    # bench.py inside a job whose world size is not --gpus: a message that names both, before any device is touched
    r = subprocess.run([sys.executable, "bench.py", "--gpus", "4"], cwd=root, env=dict(env, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0"),
                       capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "--gpus 4" in r.stderr and "2-rank" in r.stderr
    # eval.py trainer=ddp dry_run=true composes the reference's ddp trainer keys and does not launch anything
    r = subprocess.run([sys.executable, "eval.py", "task_name=inference", "ckpt_path=null", "trainer=ddp", "dry_run=true", "extras.print_config=false",
                        "data.dataset.accession_code_fillter=[CLN025]", f"paths.output_dir={tmp_path}/out"], cwd=root,
                       env=dict(env, TEST_DATA=os.path.join(root, "tests", "golden", "pdb"), CACHE_DIR=str(tmp_path / "cache"), PROJECT_ROOT=str(tmp_path)),
                       capture_output=True, text=True, timeout=180)
    assert r.returncode == 0 and "dry_run" in (r.stderr + r.stdout), r.stderr[-1500:]
