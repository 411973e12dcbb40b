"""``DiffusionLitModule``: the inference surface of the reference's LightningModule
(src/models/diffusion_module.py:51-60 constructor, :214-369 predict_step).

``predict_step`` keeps the reference's contract — hyper-parameters from the ``inference:`` block of
configs/model/diffusion.yaml, one target per batch, replicas chunked by ``replica_per_batch``, one
multi-MODEL PDB per t_delta under ``<output_dir>/<t_delta>/<accession>.pdb`` and the merged
``<output_dir>/all_delta/<accession>.pdb`` — while the loop body runs on the HIP kernels
(str2str_amd/sampler.py).  New: when ``torch.distributed`` is initialised, the WHOLE ``n_replica`` range of a
(target, t_delta) is sharded over the ranks (independent trajectories): rank r owns the contiguous
replicas ``shard_range(n_replica, r, world)``, walks the reference's chunks of ``replica_per_batch`` and samples
its intersection with each (a full 64-replica launch per rank once n_replica >= 64 x world, instead of 64/world),
and ONE collective per t_delta (RCCL gather on GPUs) brings the coordinates to rank 0; rank-major concatenation
reproduces the reference's MODEL order.  In the default ``rng_mode="host"`` every rank draws each chunk's host
noise identically (also for chunks it does not sample), so the files equal a single-GPU run sample for sample.
Training hooks are out of scope (inference-only north star).  Lightning is optional: with it installed
the class is a LightningModule, without it a plain nn.Module with the same attributes.
"""
from __future__ import annotations

import os
from types import SimpleNamespace
from typing import Any, Optional

import numpy as np
import torch

from ..common.pdb_utils import AsyncPdbWriter, atom37_to_pdb, merge_pdbfiles
from ..common.rigid_utils import Rigid
from ..sampler import (forward_backward, forward_backward_chunks, iter_forward_backward_deltas, plan_mixed_work, rank_chunk_slices,
                       sample_mixed_lengths, shard_range)

try:  # pragma: no cover - depends on the environment
    from lightning import LightningModule as _Base
except Exception:  # lightning is not a dependency of the sampling path
    _Base = torch.nn.Module


def _ns(obj):
    if obj is None or isinstance(obj, SimpleNamespace):
        return obj
    if isinstance(obj, dict):
        return SimpleNamespace(**obj)
    return obj  # OmegaConf DictConfig / namespace: attribute access already works


BACKBONE_SLOTS = 5   # compute_backbone fills atom37 slots 0..4 (N, CA, C, CB, O) and nothing else (reference all_atom.py:141-173)


def compact_backbone(atom37: torch.Tensor) -> torch.Tensor:
    """[.., N, 37, 3] -> the five slots that carry coordinates, [.., N, 5, 3] (what travels over xGMI: 7.4x less than atom37)."""
    return atom37[..., :BACKBONE_SLOTS, :].contiguous()


def expand_backbone(bb: torch.Tensor) -> torch.Tensor:
    """Inverse of ``compact_backbone``: the other 32 slots of atom37 are exact zeros in ``compute_backbone``'s output."""
    out = bb.new_zeros(tuple(bb.shape[:-2]) + (37, 3))
    out[..., :BACKBONE_SLOTS, :] = bb
    return out


def gather_replicas(atom37: torch.Tensor, total: int, group=None) -> Optional[torch.Tensor]:
    """Gather every rank's replica slice [b_r, N, 37, 3] to rank 0 in replica order (ONE collective; RCCL gathers the device tensors,
    each rank over its own xGMI link to rank 0).  The payload is the compact backbone [per, N, 5, 3]; rank 0 gets atom37 back.
    Slices follow ``shard_range`` (ceil split), so they are padded to the common size for the gather."""
    import torch.distributed as dist

    world, rank = dist.get_world_size(group), dist.get_rank(group)
    per = -(-total // world)
    bb = compact_backbone(atom37)
    if dist.get_backend(group) == "gloo":   # (CPU collectives: the multi-process tests; RCCL gathers device tensors)
        bb = bb.cpu()
    pad = bb.new_zeros((per,) + tuple(bb.shape[1:]))
    pad[: bb.shape[0]] = bb
    bufs = [torch.empty_like(pad) for _ in range(world)] if rank == 0 else None
    dist.gather(pad, bufs, dst=0, group=group)
    if rank != 0:
        return None
    parts = []
    for r, buf in enumerate(bufs):
        lo, hi = shard_range(total, r, world)
        parts.append(buf[: hi - lo])
    return expand_backbone(torch.cat(parts, dim=0))


class DiffusionLitModule(_Base):
    def __init__(self, net: torch.nn.Module, optimizer: Any = None, scheduler: Any = None, diffuser: Any = None,
                 loss: Any = None, compile: bool = False, inference: Any = None):
        super().__init__()
        self.net = net
        self.diffuser = diffuser
        hp = SimpleNamespace(optimizer=optimizer, scheduler=scheduler, loss=loss, compile=compile, inference=_ns(inference))
        if _Base is torch.nn.Module:
            self.hparams = hp
        else:  # LightningModule: hparams is a managed attribute
            self.save_hyperparameters(dict(optimizer=optimizer, scheduler=scheduler, loss=loss, compile=compile,
                                           inference=inference), logger=False)
        self.rng_mode = "host"  # reference-order host noise (fixed seed == reference noise); "device" = throughput

    def forward(self, batch):
        return self.net(batch)

    def training_step(self, *a, **k):
        raise NotImplementedError("training is outside the scope of this build (inference-only)")

    @torch.no_grad()
    def predict_step(self, batch: dict, batch_idx: int = 0) -> str:
        import torch.distributed as dist

        inf = _ns(self.hparams.inference if not isinstance(self.hparams, dict) else self.hparams["inference"])
        n_replica, replica_per_batch = int(inf.n_replica), int(inf.replica_per_batch)
        delta_range = np.around(np.arange(inf.delta_min, inf.delta_max + 1e-5, inf.delta_step), decimals=2)
        self_cond = bool(inf.self_conditioning) and bool(self.net.embedder.self_conditioning)
        output_dir = inf.output_dir
        if inf.backward_only:
            n_replica *= len(delta_range)
            delta_range = [-1.0]
        assert batch["aatype"].shape[0] == 1, "Batch size must be 1 for correct inference."
        device = next(self.net.parameters()).device
        distributed = dist.is_available() and dist.is_initialized()   # (also a 1-rank group: the same collective path at every world size)
        shard = (dist.get_rank(), dist.get_world_size()) if distributed else (0, 1)
        if self.rng_mode == "device" and not getattr(self, "_device_rng_seeded", False):
            # throughput mode draws on the device generator: decorrelate the ranks once (they all start from the same
            # default seed otherwise and would share their Brownian increments)
            torch.cuda.manual_seed(torch.initial_seed() + 7919 * shard[0])
            self._device_rng_seeded = True
        accession_code = batch["accession_code"][0]
        extra = {k: batch[k][0].detach().cpu().numpy() for k in ("aatype", "chain_index", "residue_index")}
        kw = dict(num_timesteps=inf.num_timesteps, min_t=inf.min_t, noise_scale=inf.noise_scale,
                  probability_flow=inf.probability_flow, self_conditioning=self_cond, device=device, rng=self.rng_mode)
        self.last_samples = {}   # t_delta -> atom37 [n_replica, N, 37, 3] device tensor of the last target (rank 0; programmatic callers)
        # files are written behind the sampler (a group of t_deltas is gathered and queued as soon as it is sampled; the GPU goes on
        # with the next group meanwhile -- groups of ONE t_delta for targets too large to merge); the context manager stops the worker
        # and frees its page-locked buffers on every path -- after an exception the files still queued are dropped, not written
        with AsyncPdbWriter() as writer:
            gt4 = batch["rigidgroups_gt_frames"][..., 0, :, :].clone()
            # The reference's chunks (and its loop over t_delta) are units of its host noise stream, not of the arithmetic: the chunks --
            # and the t_deltas -- of a target whose replicas fit a pair budget are sampled as ONE batch (sampler.iter_forward_backward_deltas:
            # same samples file for file, far fewer launches; S2S_MERGE_DELTAS=0 = one t_delta at a time, S2S_MERGE_CHUNKS=0 = one
            # trajectory per chunk); an empty slice still advances the host generator in lock-step with the other ranks
            groups = iter_forward_backward_deltas(self.net, self.diffuser, batch, gt4, rank_chunk_slices(n_replica, replica_per_batch, *shard),
                                                  [float(t) for t in delta_range], **kw)
            for idx, samples in groups:   # a group's gather + files run while the GPU is already sampling the next group
                for i, a37 in zip(idx, samples):
                    t_delta = delta_range[i]
                    if distributed:
                        a37 = gather_replicas(a37, n_replica)   # ONE collective per (target, t_delta)
                    if shard[0] == 0:
                        self.last_samples[float(t_delta)] = a37
                        t_dir = os.path.join(output_dir, f"{t_delta}")
                        os.makedirs(t_dir, exist_ok=True)
                        writer.submit(a37, os.path.join(t_dir, f"{accession_code}.pdb"), **extra)
            saved = writer.results()
        all_dir = os.path.join(output_dir, "all_delta")
        if shard[0] == 0:
            os.makedirs(all_dir, exist_ok=True)
            merge_pdbfiles(saved, os.path.join(all_dir, f"{accession_code}.pdb"), verbose=False)
        if distributed:
            dist.barrier()
        return all_dir

    @torch.no_grad()
    def predict_mixed(self, batches) -> str:
        """ALL targets of a run in length-bucketed, padded, masked batches (BASELINE configs[4]) instead of one ``predict_step`` per
        target -- ``model.inference.mixed_batch=true``.  The reference cannot do this (it asserts one target per batch,
        diffusion_module.py:249, and its padding semantics would be wrong for it); here padding is exact (sampler.
        sample_mixed_lengths), so every chain is sampled as if it ran alone.  Same hyper-parameters, same output tree
        (``<output_dir>/<t_delta>/<accession>.pdb`` + ``all_delta``).  The (chain, replica block) items are distributed over the
        ranks by FLOP weight (``plan_mixed_work``); ONE gather of a flat buffer of compact backbones [., 5, 3] per t_delta brings them to rank 0.  The
        noise stream is per (chain, block) -- not the reference's per-target chunk order -- so this is the throughput mode: same
        distribution, different draws than ``predict_step`` under the same seed."""
        import torch.distributed as dist

        inf = _ns(self.hparams.inference if not isinstance(self.hparams, dict) else self.hparams["inference"])
        n_replica = int(inf.n_replica)
        delta_range = np.around(np.arange(inf.delta_min, inf.delta_max + 1e-5, inf.delta_step), decimals=2)
        if inf.backward_only:
            n_replica *= len(delta_range)
            delta_range = [-1.0]
        self_cond = bool(inf.self_conditioning) and bool(self.net.embedder.self_conditioning)
        device = next(self.net.parameters()).device
        distributed = dist.is_available() and dist.is_initialized()
        rank, world = (dist.get_rank(), dist.get_world_size()) if distributed else (0, 1)
        if self.rng_mode == "device" and not getattr(self, "_device_rng_seeded", False):
            torch.cuda.manual_seed(torch.initial_seed() + 7919 * rank)
            self._device_rng_seeded = True
        targets = list(batches)
        for tg in targets:
            assert tg["aatype"].shape[0] == 1, "one chain per dataloader batch"
        lens = [int(tg["aatype"].shape[1]) for tg in targets]
        plan = plan_mixed_work(lens, n_replica, world)
        seed_base = int(torch.initial_seed())   # the run seed (eval.py seeds every rank alike); see sampler.mixed_batch_seed
        saved = {k: [] for k in range(len(targets))}
        self.last_samples = {}
        for t_delta in delta_range:
            pieces = sample_mixed_lengths(self.net, self.diffuser, targets, n_replica, float(t_delta), num_timesteps=inf.num_timesteps,
                                          min_t=inf.min_t, noise_scale=inf.noise_scale, probability_flow=inf.probability_flow,
                                          self_conditioning=self_cond, device=device, shard=(rank, world), rng=self.rng_mode, plan=plan,
                                          seed_base=seed_base)
            # this rank's pieces in plan order (chain, replica_lo, replica_hi) -> one flat buffer
            mine = [(k, lo, p) for k, ps in enumerate(pieces) for lo, p in ps]
            flat = torch.cat([compact_backbone(p).reshape(-1) for _, _, p in mine]) if (mine and distributed) else torch.zeros(0, device=device)
            per_chain = {k: [] for k in range(len(targets))}
            if distributed:
                cpu = dist.get_backend() == "gloo"
                buf_dev = torch.device("cpu") if cpu else device
                # every rank can compute every rank's piece list from the plan: sizes are known, only the payload travels
                layout = [sorted(((k, lo, hi) for b in plan[r] for k, lo, hi in b["items"]), key=lambda x: (x[0], x[1])) for r in range(world)]
                sizes = [sum((hi - lo) * lens[k] * BACKBONE_SLOTS * 3 for k, lo, hi in layout[r]) for r in range(world)]   # compact [., 5, 3]
                pad = torch.zeros(max(sizes), device=buf_dev)
                pad[: flat.numel()] = flat.to(buf_dev)
                bufs = [torch.empty_like(pad) for _ in range(world)] if rank == 0 else None
                dist.gather(pad, bufs, dst=0)    # ONE collective per t_delta
                if rank == 0:
                    for r in range(world):
                        o = 0
                        for k, lo, hi in layout[r]:
                            n = (hi - lo) * lens[k] * BACKBONE_SLOTS * 3
                            per_chain[k].append((lo, expand_backbone(bufs[r][o:o + n].view(hi - lo, lens[k], BACKBONE_SLOTS, 3))))
                            o += n
            else:
                for k, lo, p in mine:
                    per_chain[k].append((lo, p))
            if rank == 0:
                t_dir = os.path.join(inf.output_dir, f"{t_delta}")
                os.makedirs(t_dir, exist_ok=True)
                for k, tg in enumerate(targets):
                    a37 = torch.cat([p for _, p in sorted(per_chain[k], key=lambda x: x[0])], dim=0)
                    assert a37.shape[0] == n_replica, (a37.shape, n_replica)
                    code = tg["accession_code"][0]
                    self.last_samples[(code, float(t_delta))] = a37
                    extra = {kk: tg[kk][0].detach().cpu().numpy() for kk in ("aatype", "chain_index", "residue_index")}
                    saved[k].append(atom37_to_pdb(atom_positions=a37.cpu().numpy(), save_to=os.path.join(t_dir, f"{code}.pdb"), **extra))
        all_dir = os.path.join(inf.output_dir, "all_delta")
        if rank == 0:
            os.makedirs(all_dir, exist_ok=True)
            for k, tg in enumerate(targets):
                merge_pdbfiles(saved[k], os.path.join(all_dir, f"{tg['accession_code'][0]}.pdb"), verbose=False)
        if distributed:
            dist.barrier()
        return all_dir
