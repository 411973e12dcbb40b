"""The N>1 path on CPU: world_size-2 gloo processes exercise the replica sharding + ONE gather per chunk
that DiffusionLitModule.predict_step / bench.py use on RCCL (sampling itself needs the GPU)."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, total, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from str2str_amd.models.diffusion_module import gather_replicas
    from str2str_amd.sampler import shard_range

    lo, hi = shard_range(total, rank, world)
    # stand-in for this rank's sampled coordinates: replica r is filled with the value r
    mine = torch.arange(lo, hi, dtype=torch.float32)[:, None, None, None].expand(hi - lo, 4, 37, 3).contiguous()
    out = gather_replicas(mine, total)
    if rank == 0:
        q.put(out[:, 0, 0, 0].tolist())
    else:
        assert out is None
    # identical host noise on every rank (same seed) -> slices of one stream
    torch.manual_seed(7)
    z = torch.randn(total, 3)
    gathered = [torch.empty_like(z) for _ in range(world)]
    dist.all_gather(gathered, z)
    assert all(torch.equal(g, z) for g in gathered)
    dist.barrier()
    dist.destroy_process_group()


def _run(total):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, total, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = q.get(timeout=120)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    return got


def test_gather_keeps_replica_order_even_split():
    assert _run(6) == [0.0, 1.0, 2.0, 3.0, 4.0, 5.0]


def test_gather_keeps_replica_order_uneven_split():
    assert _run(5) == [0.0, 1.0, 2.0, 3.0, 4.0]
