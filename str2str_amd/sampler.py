"""The forward-backward diffusion sampler (device-resident loop).

This is the closure ``forward_backward`` of the reference's ``DiffusionLitModule.predict_step``
(src/models/diffusion_module.py:260-334) with the same schedule semantics:
``n = int(num_timesteps*T)``, ``dt = 1/n`` (not the spacing of ``ts``), ``ts = linspace(min_t, T, n)[::-1]``,
one extra network evaluation for self-conditioning, and the last step (``t == min_t``) returning the
x0 prediction itself.  Differences in HOW:
  * all per-step scalars (sigma bin, g(t)^2, exp(-beta/2), ...) are computed on the host once per
    trajectory; the loop body is network forward -> ONE fused SE(3) kernel, with no device->host sync;
  * the backbone projection runs once at the end (the reference recomputes it every forward and
    throws the result away, denoising_ipa.py:197-201);
  * the default arithmetic (split-f16 matrix products, str2str_amd/arith.py) has f16's RANGE: its kernels raise a device flag
    when a value they split reaches 2^15; the flag is read once per chunk, where the loop synchronises anyway, and a flagged
    chunk is re-run from its starting frames (same noise) on the exact fp32 kernels -- after which the network stays there;
  * replicas can be sharded over ranks (``shard=(rank, world)``): noise for the WHOLE chunk is drawn
    on every rank from the same host generator state and sliced, so the union over ranks equals the
    single-process result sample for sample ("parity" RNG mode).  ``rng="device"`` instead draws
    nothing on the host inside the loop (throughput mode; different noise stream).
"""
from __future__ import annotations

import logging
import os
from typing import Optional, Tuple

import numpy as np
import torch

from . import ops
from .arith import FAMILIES, family_slots, net_arith, use_arith
from .common.all_atom import compute_backbone
from .common.rigid_utils import Rigid

_REPEAT_KEYS = ("aatype", "residue_mask", "fixed_mask", "residue_idx", "torsion_angles_sin_cos")
_log = logging.getLogger("str2str_amd.sampler")
# A network evaluation is ~165 kernel launches (profiles/r02m_eval_sequence.md).  For tiny chunks (<= 64 Ki pairs, e.g. 64 replicas
# of a 20-residue peptide) the GPU finishes them faster than the host can issue them, so the evaluation is captured once into a
# HIP graph and replayed every step: 1.7x there, bit-identical output; larger chunks are GPU-bound (the host already runs ahead)
# and capture would only add its three warm-up evaluations (measured 0.92-0.97x).  S2S_HIP_GRAPH=0 / 1 forces eager / graph.
_GRAPH_MAX_PAIRS = 1 << 16
_GRAPH_MIN_STEPS = 16


class _GraphedNet:
    """One captured network evaluation on static input buffers (rigids_t, sc_ca_t, t_emb); outputs are static too."""

    def __init__(self, net, feats):
        self.feats = dict(feats)
        self.static = tuple(k for k in ("rigids_t", "sc_ca_t", "t_emb", "t_img") if k in feats)
        for k in self.static:
            self.feats[k] = feats[k].clone()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):  # warm-up on a side stream (caches, BLAS workspaces), as graph capture requires
            for _ in range(2):
                _net_eval(net, self.feats, True)
        torch.cuda.current_stream().wait_stream(side)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.out = _net_eval(net, self.feats, True)
        # The captured kernels hold raw pointers to every tensor the evaluation read.  Those allocated INSIDE the capture live in the
        # graph's private pool; the ones built by the warm-up outside it must be kept alive by THIS object, because their only other
        # owner is a single-slot cache that the next evaluation of another shape overwrites: the embedder's per-target tables
        # (relative-position table, its column-blocked copy, positional node image, device residue indices) and the packed /
        # derived weights of the modules' parameter caches.
        self._pinned = _graph_read_tensors(net)

    def __call__(self, feats):
        for k in self.static:
            self.feats[k].copy_(feats[k])
        self.graph.replay()
        return self.out


def _graph_read_tensors(net):
    """Every tensor held by a replaceable cache slot of the network's modules (index tables of the embedder, ParamCache entries)."""
    from .models.net.layers import ParamCache

    def tensors(x, out):
        if torch.is_tensor(x):
            out.append(x)
        elif isinstance(x, dict):
            for v in x.values():
                tensors(v, out)
        elif isinstance(x, (list, tuple)):
            for v in x:
                tensors(v, out)
        return out

    keep = []
    for m in net.modules():
        for name in ("_idx_val", "_rel_cb", "_idx_src", "_fx_val", "_fx_src", "_mk_val", "_mk_src", "node_embed_act"):
            if hasattr(m, name):
                tensors(getattr(m, name), keep)
        for v in vars(m).values():
            if isinstance(v, ParamCache):
                tensors(v.pinned(), keep)
    keep += list(ops._CONST_ROWS.values())   # constant pre-scale rows of the node layers (a bounded cache that may drop them later)
    return keep


def _net_eval(net, feats, defer_psi: bool):
    """One evaluation inside the loop: the psi blend is deferred to the last step where the network supports it."""
    if defer_psi and hasattr(net, "blend_psi"):
        return net(feats, as_tensor_7=False, defer_psi_blend=True)
    return net(feats, as_tensor_7=False)


_GRAPH_CACHE = {}   # (net id, parameter versions, b, N, feature bytes) -> _GraphedNet; a few entries (one per chunk shape)
_GRAPH_CACHE_MAX = 4


def _graph_key(net, feats, b, N):
    import hashlib

    h = hashlib.sha1()
    for k in sorted(feats):
        v = feats[k]
        if k in ("rigids_t", "sc_ca_t", "t_emb", "t_img", "t") or not torch.is_tensor(v):
            continue  # per-step inputs are copied into the static buffers at every replay
        h.update(k.encode()); h.update(str(tuple(v.shape)).encode()); h.update(v.detach().cpu().numpy().tobytes())
    tr = getattr(net, "translator", None)
    # which kernels were captured: the arithmetic / kernel selectors of the modules
    modes = tuple(str(getattr(m, a)) for m, a in family_slots(net))   # per switch, in module order: a mixed network has many forms
    modes += tuple(int(m.prescale_exp) for m in net.modules() if hasattr(m, "prescale_exp"))   # block exponents are launch arguments
    modes += tuple(bool(m.fold) for m in net.modules() if hasattr(m, "fold"))                 # folded / per-head attention operands (A/B switch)
    return (id(net), sum(p._version for p in net.parameters()), b, N, bool(getattr(tr, "exact_padding", False)),
            bool(getattr(tr, "fuse_pair_projection", False)), modes, h.hexdigest())


def _maybe_graph(net, feats, b, N, trace, n_steps):
    mode = os.environ.get("S2S_HIP_GRAPH", "auto")
    if mode == "0" or trace is not None or ops.KernelTimer.active is not None or "t_emb" not in feats:
        return None
    if mode != "1" and (b * N * N > _GRAPH_MAX_PAIRS or n_steps < _GRAPH_MIN_STEPS):
        return None
    key = _graph_key(net, feats, b, N)   # chunks / t_deltas of one target share the capture (same static features)
    hit = _GRAPH_CACHE.get(key)
    if hit is not None:
        return hit
    try:
        g = _GraphedNet(net, feats)
    except ops.HipLibraryError:
        raise                            # a kernel / ABI error is an error, not a capture problem
    except RuntimeError as e:            # capture itself is an optimisation: fall back to eager launches
        _log.warning("HIP graph capture failed (%s); continuing with eager launches", repr(e)[:200])
        torch.cuda.synchronize()
        return None
    if len(_GRAPH_CACHE) >= _GRAPH_CACHE_MAX:
        _GRAPH_CACHE.pop(next(iter(_GRAPH_CACHE)))
    _GRAPH_CACHE[key] = g
    return g


def schedule(t_delta: float, num_timesteps: int, min_t: float):
    T = t_delta if t_delta > 0 else 1.0
    n = int(float(num_timesteps) * T)
    return T, n, 1.0 / n, np.linspace(min_t, T, n)[::-1]


def shard_range(total: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous replica range of ``rank``: ceil split, rank-major order == replica order."""
    per = -(-total // world)
    lo = min(total, rank * per)
    return lo, min(total, lo + per)


def rank_chunk_slices(n_replica: int, replica_per_batch: int, rank: int, world: int):
    """How ``rank`` walks the reference's replica chunks (diffusion_module.py:341-351: ``n_replica`` split into chunks of
    ``replica_per_batch``, the unit of its host noise stream) when the WHOLE replica range is sharded over ``world`` ranks:
    -> [(chunk_size, lo, hi)] per chunk, (lo, hi) = the part of that chunk inside this rank's contiguous range
    ``shard_range(n_replica, rank, world)`` ((0, 0) when the chunk lies outside it).  Rank-major concatenation of the
    sampled slices is the single-process replica order."""
    my_lo, my_hi = shard_range(n_replica, rank, world)
    sizes = [replica_per_batch] * (n_replica // replica_per_batch)
    if n_replica % replica_per_batch > 0:
        sizes.append(n_replica % replica_per_batch)
    out, c0 = [], 0
    for bsz in sizes:
        lo, hi = max(my_lo, c0) - c0, min(my_hi, c0 + bsz) - c0
        out.append((bsz, lo, hi) if hi > lo else (bsz, 0, 0))
        c0 += bsz
    return out


@torch.no_grad()
def denoise_loop(net, diffuser, feats: dict, rigids_t: torch.Tensor, ts, dt: float, *, min_t: float,
                 noise_scale: float = 1.0, probability_flow: bool = True, self_conditioning: bool = True,
                 center_mode: int = 1, host_noise=None, trace: Optional[list] = None):
    """The loop body of forward_backward over ALREADY EXPANDED per-sample features (every tensor has the
    sample dimension b) starting from the noised frames ``rigids_t`` [b,N,7]: 1 self-conditioning evaluation,
    then per step  network -> fused SE(3) step, last step returns the x0 prediction.  ``host_noise()`` (optional)
    returns the (z_rot, z_trans) float64 [b,N,3] device tensors of a step or None.  -> (atom37, final rigids7, psi).

    Range guard: in the split-f16 arithmetic the pass runs with the library's range flag cleared; if a kernel raised it (an
    activation reached 2^15 -- f16 tops out at 65504) or the result is not finite, the SAME chunk (same starting frames, same
    noise) is run again with ONLY the kernel families that raised it (str2str_amd/arith.py: node stream, edge transition, edge
    embedding, IPA) on their exact fp32 kernels; a warning is logged once per family and those families stay in fp32 for later
    chunks (``net.range_fallback`` = the set of demoted families) -- one out-of-range activation in, say, the encoder attention no
    longer costs the edge transitions (73 % of the step) their 3x.  One flag read per chunk, where the loop synchronises anyway.
    The edge transitions -- the family whose demotion would cost 2x of the whole step -- are first given a BLOCK EXPONENT instead
    (``EdgeTransition.prescale_exp`` 0 -> 5 -> 10 -> 15: the kernel keeps its hidden activations as planes of 2^-e x the value, exact
    and at the same speed, include/str2str_hip.h) and only go to fp32 when 2^30 is not enough; ``net.range_prescale`` records it."""
    kw = dict(min_t=min_t, noise_scale=noise_scale, probability_flow=probability_flow, self_conditioning=self_conditioning,
              center_mode=center_mode, trace=trace)
    return _range_guarded(net, lambda: _denoise_pass(net, diffuser, feats, rigids_t, ts, dt, host_noise=host_noise, **kw),
                          rigids_t.device, host_draws=host_noise is not None, device_draws=host_noise is None and not probability_flow,
                          trace=trace)


def _range_guarded(net, run_pass, device, *, host_draws: bool, device_draws: bool, trace: Optional[list] = None):
    """``run_pass()`` (one complete pass over a batch of trajectories from their starting frames -> a tuple whose element 1 holds the
    final frames) under the range guard described in ``denoise_loop``: repeated with the flagged kernel families demoted (or the
    edge transitions' block exponent raised) until no flag is raised; every repetition sees the same noise (generator states
    restored)."""
    if net_arith(net) == "f32":
        out = run_pass()
        _require_finite(out[1], "fp32")
        return out
    demoted = set(getattr(net, "range_fallback", None) or ())
    rng_state = torch.cuda.get_rng_state(device) if device_draws else None
    # Host noise (parity mode) comes from the global CPU generator (forward_backward.host_noise): a replay re-draws it from the
    # generator state saved here -- nothing is recorded (the draws of a long SDE trajectory at b = 128, N = 512 are ~0.6 GB), and
    # every pass leaves the generator where one completed pass leaves it.
    host_state = torch.get_rng_state() if host_draws else None
    while True:
        if host_state is not None:
            torch.set_rng_state(host_state)
        ops.range_flag_reset()
        try:
            with use_arith(net, "f32", families=tuple(demoted)):
                out = run_pass()
            finite = bool(torch.isfinite(out[1]).all())     # (the synchronisation point of the chunk)
            bits = ops.range_flag_read()
            if bits == 0 and finite:
                return out
            new = set(ops.range_families(bits)) - demoted
            why = f"an activation left f16's safe range in the split-f16 kernels ({ops.range_flag_names(bits)}{'' if finite else '; non-finite frames'})"
            if "edge_transition" in new:
                e_now = _raise_edge_prescale(net)
                if e_now:
                    new.discard("edge_transition")
                    _log.warning("range guard: %s; the edge transitions keep the split-f16 kernel with a block exponent of %d on their "
                                 "hidden activations (planes of 2^-%d x the value: exact, same speed; usable range 2^%d) -- re-running this chunk",
                                 why, e_now, e_now, 15 + e_now)
                    net.range_prescale = {"edge_transition": e_now}
                    if not new:
                        if trace is not None:
                            del trace[:]
                        if rng_state is not None:
                            torch.cuda.set_rng_state(rng_state, device)
                        continue
        except ops.WeightRangeError as e:   # |32 w| >= 65504: the weights themselves cannot be packed for the f16 kernels
            new = set(FAMILIES) - demoted
            if not new:     # every family already runs its exact fp32 kernels (which pack nothing): not a range event, the caller's error
                raise
            why = f"a weight does not fit the split-f16 packing ({e})"
        if not new:
            if len(demoted) == len(FAMILIES):
                _require_finite(out[1], "fp32 (range fallback)")
                return out
            new = set(FAMILIES) - demoted   # non-finite frames without a flag: nothing names a family, everything goes exact
        demoted |= new
        _log.warning("range guard: %s; re-running this chunk with the %s kernels on exact fp32 and keeping them there (now in fp32: %s; "
                     "S2S_ARITH=f32 starts everything in fp32)", why, " + ".join(sorted(new)), ", ".join(sorted(demoted)))
        net.range_fallback = frozenset(demoted)
        if trace is not None:
            del trace[:]
        if rng_state is not None:
            torch.cuda.set_rng_state(rng_state, device)


def _raise_edge_prescale(net, step: int = 5, top: int = 15) -> int:
    """Next block exponent of the network's EdgeTransition kernels (all of them: the guard's flag names the family); 0 = none left."""
    mods = [m for m in net.modules() if hasattr(m, "prescale_exp") and getattr(m, "arith", None) == "f16x3"]
    cur = max((int(m.prescale_exp) for m in mods), default=top)
    if not mods or cur >= top:
        return 0
    for m in mods:
        m.prescale_exp = min(cur + step, top)
    return min(cur + step, top)


def _require_finite(rigids7, what):
    if not bool(torch.isfinite(rigids7).all()):
        raise ops.HipLibraryError(f"non-finite frames at the end of the trajectory in the {what} arithmetic: the network itself "
                                  "overflows fp32 on this input (weights / checkpoint?)")


def _denoise_pass(net, diffuser, feats: dict, rigids_t: torch.Tensor, ts, dt: float, *, min_t: float, noise_scale: float,
                  probability_flow: bool, self_conditioning: bool, center_mode: int, host_noise, trace: Optional[list]):
    """One pass of the loop in the network's current arithmetic (see ``denoise_loop``)."""
    device = rigids_t.device
    b, N = rigids_t.shape[:2]
    feats = dict(feats)
    # the masks as float32 device tensors ONCE per chunk: the network's per-chunk caches (embedder terms, mask terms) key on their identity
    for k in ("residue_mask", "fixed_mask"):
        feats[k] = feats[k].to(device).float().contiguous()
    mask = feats["residue_mask"]
    diffuse_mask = ((1 - feats["fixed_mask"]) * mask).contiguous()
    t_all = torch.as_tensor(np.ascontiguousarray(ts, dtype=np.float64)).float()  # fl32(t), as `t * torch.ones(B)` gives
    p8_all = diffuser.step_params(t_all).to(device)  # [n, 8]: t is uniform over the chunk
    # timestep embeddings of the whole schedule, uploaded once (same host function the network would call per step)
    temb_all = net.embedder.time_embed(t_all).to(device) if hasattr(getattr(net, "embedder", None), "time_embed") else None
    # ... and their first-layer images (the embedder's only t-dependent arithmetic), for the whole schedule at once
    timg_all = net.embedder.time_images(temb_all) if temb_all is not None and hasattr(net.embedder, "time_images") else None
    keep_bb = getattr(net, "backbone_in_forward", None)
    if keep_bb is not None:
        net.backbone_in_forward = False
    try:
        feats["rigids_t"] = rigids_t
        feats["sc_ca_t"] = torch.zeros(b, N, 3, device=device)
        feats["t"] = torch.full((b,), float(ts[0]), dtype=torch.float32)
        if temb_all is not None:
            feats["t_emb"] = temb_all[0]
        if timg_all is not None:
            feats["t_img"] = timg_all[0]
        graphed = _maybe_graph(net, feats, b, N, trace, len(ts))
        run = graphed if graphed is not None else (lambda f: _net_eval(net, f, trace is None))
        if self_conditioning:
            feats["sc_ca_t"] = run(feats)["rigids7"][..., 4:].clone()
        final = None
        for k, t in enumerate(ts):
            feats["t"] = torch.full((b,), float(t), dtype=torch.float32)
            if temb_all is not None:
                feats["t_emb"] = temb_all[k]
            if timg_all is not None:
                feats["t_img"] = timg_all[k]
            out = run(feats)
            x0_7 = out["rigids7"]
            if t == min_t:
                final = out
                if trace is not None:
                    trace.append(dict(t=t, rigids_t=feats["rigids_t"], sc_ca_t=feats["sc_ca_t"], x0=x0_7, psi=out["psi"]))
                break
            sc_in = feats["sc_ca_t"]
            if self_conditioning:
                feats["sc_ca_t"] = x0_7[..., 4:].clone() if graphed is not None else x0_7[..., 4:]
            z = host_noise() if host_noise is not None else None
            z_rot, z_trans = z if z is not None else (None, None)
            if not probability_flow and z_rot is None:
                z_rot = torch.randn(b, N, 3, dtype=torch.float64, device=device)
                z_trans = torch.randn(b, N, 3, dtype=torch.float64, device=device)
            p8 = p8_all[k].expand(b, 8).contiguous()
            nxt, rs, tsc = diffuser.step(x0_7, feats["rigids_t"], p8, dt, mask, diffuse_mask, center_trans=center_mode,
                                         noise_scale=noise_scale, probability_flow=probability_flow, z_rot=z_rot,
                                         z_trans=z_trans, want_scores=trace is not None)
            if trace is not None:
                trace.append(dict(t=t, rigids_t=feats["rigids_t"], sc_ca_t=sc_in, x0=x0_7, psi=out["psi"], rot_score=rs,
                                  trans_score=tsc, next7=nxt))
            feats["rigids_t"] = nxt
        if final.get("psi_deferred"):   # the blend with the input torsion under the fixed mask, once (DenoisingNet.blend_psi)
            final = dict(final, psi=net.blend_psi(final["psi"], feats["torsion_angles_sin_cos"], feats["fixed_mask"]))
        atom37 = compute_backbone(final["rigids"], final["psi"], aatype=feats.get("aatype"), _rigids7=final["rigids7"])[0]
    finally:
        if keep_bb is not None:
            net.backbone_in_forward = keep_bb
    return atom37, final["rigids7"], final["psi"]


def _require_hip_device(device, net) -> torch.device:
    device = torch.device(device) if device is not None else next(net.parameters()).device
    if device.type != "cuda":
        raise ops.HipLibraryError("the sampler needs the HIP device (MI355X); there is no CPU fallback")
    return device


def _start_frames(diffuser, batch: dict, rigids_0: Rigid, t_delta: float, lo: int, hi: int, rng: str, device):
    """The noised frames a chunk's trajectory starts from, for the replicas [lo, hi) of the chunk ``rigids_0`` -> [hi - lo, N, 7]
    float32 device tensor, or None for an empty slice.  ``rng="host"``: drawn once per trajectory on the host generators for the
    WHOLE chunk, in the reference's order (diffusion_module.py:277-296 there) -- also for an empty slice; ``rng="device"``
    (throughput mode): for the slice only, drawn and applied on the device (s2s_forward_marginal)."""
    B_total, N = rigids_0.shape[0], rigids_0.shape[1]
    b = hi - lo
    if rng == "device":
        if b == 0:
            return None
        if t_delta > 0:
            dmask = batch["residue_mask"].to(device).float().reshape(1, N).expand(b, N).contiguous()
            return diffuser.forward_marginal_device(rigids_0[lo:hi].to_tensor_4x4().to(device), t_delta, dmask)
        return diffuser.forward_marginal_device(None, None, shape=(b, N))
    # ~300 small host tensor operations on [B, N, 3..9] values: with every core of the host in torch's intra-op pool each of them pays
    # the pool's wake-up (tens of ms per chunk on a 128-thread box, in front of the GPU work); element-wise arithmetic and reductions
    # over the last axis do not depend on the thread count, so the frames are the same bits.  S2S_HOST_FM_THREADS=0: leave the pool alone.
    nt, keep = int(os.environ.get("S2S_HOST_FM_THREADS", "1")), torch.get_num_threads()
    if nt > 0 and nt != keep:
        torch.set_num_threads(nt)
    try:
        if t_delta > 0:
            rigids_t = diffuser.forward_marginal(rigids_0=rigids_0.to(device="cpu"), t=t_delta * torch.ones(B_total),
                                                 diffuse_mask=batch["residue_mask"].cpu().repeat(B_total, 1),
                                                 as_tensor_7=True)["rigids_t"]
        else:
            rigids_t = diffuser.sample_prior(shape=rigids_0.shape, device="cpu", as_tensor_7=True)["rigids_t"]
    finally:
        if nt > 0 and nt != keep:
            torch.set_num_threads(keep)
    if b == 0:
        return None
    return rigids_t[lo:hi].to(device).float().contiguous()


def _burn_step_draws(B_total: int, N: int, n_draw_steps: int):
    """The reference consumes two float64 normal draws of the whole chunk per step even under the probability-flow ODE
    (so3.py:360, r3.py:109): consume them so that the host generator is where the reference's is for the next chunk.  Nothing reads the
    values, so the generator is fast-forwarded over them where that is exact (ops.host_rng_discard_float64_normals: the engine's
    position after the draws, checked against real draws once per process); otherwise they are drawn."""
    if ops.host_rng_discard_float64_normals(B_total * N * 3, 2 * n_draw_steps):
        return
    for _ in range(n_draw_steps):
        torch.randn(B_total, N, 3, dtype=torch.float64)
        torch.randn(B_total, N, 3, dtype=torch.float64)


def _skips_unused_draws(probability_flow: bool, B_total: int, N: int) -> bool:
    """Under the probability-flow ODE the reference's per-step draws are consumed, not used.  Where the host generator can be
    fast-forwarded over them (``_burn_step_draws``: milliseconds per chunk) that happens right behind the chunk's start frames;
    otherwise they are drawn step by step inside the loop, behind the GPU work (``host_noise`` of the callers)."""
    return probability_flow and ops.host_rng_can_discard(B_total * N * 3)


@torch.no_grad()
def forward_backward(net, diffuser, batch: dict, rigids_0: Rigid, t_delta: float, *, num_timesteps: int,
                     min_t: float = 0.01, noise_scale: float = 1.0, probability_flow: bool = True,
                     self_conditioning: bool = True, device=None, shard: Tuple[int, int] = (0, 1),
                     replica_slice: Optional[Tuple[int, int]] = None, rng: str = "host", trace: Optional[list] = None,
                     return_rigids: bool = False):
    """-> atom37 [b, N, 37, 3] float32 DEVICE tensor for this rank's replica slice (b = hi - lo).

    The chunk (``rigids_0.shape[0]`` replicas) is the unit of the reference's host noise stream.  ``replica_slice`` =
    (lo, hi) selects the replicas of the chunk THIS process samples (``shard`` = (rank, world) is the ceil split of
    the chunk); in ``rng="host"`` mode every process still draws the whole chunk's noise in the reference's order --
    including a process whose slice is empty -- so the union over processes equals the single-process run sample for
    sample and every generator stays in lock-step for later chunks.  ``rng="device"`` (throughput mode) draws the
    forward-marginal and step noise on the device generator for the slice only."""
    device = _require_hip_device(device, net)
    B_total = rigids_0.shape[0]
    lo, hi = replica_slice if replica_slice is not None else shard_range(B_total, *shard)
    if not (0 <= lo <= hi <= B_total):
        raise ValueError(f"replica_slice {(lo, hi)} outside the chunk of {B_total} replicas")
    b = hi - lo
    T, n, dt, ts = schedule(t_delta, num_timesteps, min_t)
    N = rigids_0.shape[1]

    rigids_t = _start_frames(diffuser, batch, rigids_0, t_delta, lo, hi, rng, device)
    skip_ahead = rng == "host" and (rigids_t is None or _skips_unused_draws(probability_flow, B_total, N))
    if skip_ahead:      # nothing reads the chunk's step draws: the generator goes straight to where they leave it
        _burn_step_draws(B_total, N, len(ts) - 1)
    if rigids_t is None:
        return torch.zeros(0, N, 37, 3, device=device)
    feats = {k: batch[k].to(device).repeat(b, *(1,) * (batch[k].ndim - 1)) for k in _REPEAT_KEYS if k in batch}

    def host_noise():
        # the reference consumes two float64 normal draws per step even under the probability-flow ODE
        # (so3.py:360, r3.py:109): keep the host generator in lock-step for later chunks
        zr = torch.randn(B_total, N, 3, dtype=torch.float64)
        zt = torch.randn(B_total, N, 3, dtype=torch.float64)
        if probability_flow:
            return None
        return zr[lo:hi].to(device).contiguous(), zt[lo:hi].to(device).contiguous()

    atom37, r7, psi = denoise_loop(net, diffuser, feats, rigids_t, ts, dt, min_t=min_t, noise_scale=noise_scale,
                                   probability_flow=probability_flow, self_conditioning=self_conditioning, center_mode=1,
                                   host_noise=host_noise if rng == "host" and not skip_ahead else None, trace=trace)
    if return_rigids:
        return atom37, r7, psi
    return atom37


_MERGE_MAX_PAIRS = 8 << 20   # a merged trajectory stays within the working set of BASELINE configs[1] (128 x 256^2 pairs, ~11 GB)


def merge_chunk_groups(chunks, N: int, *, mergeable: bool = True, max_pairs: int = None):
    """Consecutive replica chunks ``[(chunk_size, lo, hi)]`` (``rank_chunk_slices``) -> groups that are sampled as ONE trajectory each.
    A group grows while this rank's replicas in it stay below ``max_pairs`` pairs; ``S2S_MERGE_CHUNKS=0`` (or ``mergeable=False``:
    host noise under the SDE, where a chunk's per-step draws interleave with its trajectory) keeps one chunk per group."""
    max_pairs = _MERGE_MAX_PAIRS if max_pairs is None else max_pairs
    if not mergeable or os.environ.get("S2S_MERGE_CHUNKS", "1") == "0":
        return [[c] for c in chunks]
    groups, cur, cur_b = [], [], 0
    for c in chunks:
        b = c[2] - c[1]
        if cur and (cur_b + b) * N * N > max_pairs:
            groups.append(cur)
            cur, cur_b = [], 0
        cur.append(c)
        cur_b += b
    if cur:
        groups.append(cur)
    return groups


@torch.no_grad()
def forward_backward_chunks(net, diffuser, batch: dict, gt_frames_4x4: torch.Tensor, chunks, t_delta: float, *, num_timesteps: int,
                            min_t: float = 0.01, noise_scale: float = 1.0, probability_flow: bool = True,
                            self_conditioning: bool = True, device=None, rng: str = "host", max_pairs: int = None):
    """All replica chunks of one (target, t_delta) -> atom37 [sum(hi - lo), N, 37, 3] in replica order.

    The reference samples ``n_replica`` replicas in chunks of ``replica_per_batch`` (diffusion_module.py:341-351), one trajectory per
    chunk; its default block (100 replicas in chunks of 64 + 36 on chains of 35 .. 80 residues) leaves the GPU launch-bound -- an
    evaluation of 64 x 35^2 pairs costs what one of 100 x 35^2 costs.  The chunk is only the unit of the reference's HOST NOISE
    stream: every replica's trajectory is independent of its batch (the kernels are batch-invariant: sharded == single is tested
    bit for bit).  So the chunks' start frames are drawn chunk by chunk in the reference's order (and, in ``rng="host"`` mode, the
    two float64 draws per step the reference consumes even under the ODE are consumed per chunk), and the chunks that fit a pair
    budget run as ONE trajectory: same samples, file for file, in fewer launches.  Under the SDE with host noise the per-step draws
    are part of the trajectory: one chunk per trajectory, as before.  ``chunks`` = ``rank_chunk_slices(...)`` of this rank."""
    device = _require_hip_device(device, net)
    N = gt_frames_4x4.shape[-3]
    T, n, dt, ts = schedule(t_delta, num_timesteps, min_t)
    kw = dict(num_timesteps=num_timesteps, min_t=min_t, noise_scale=noise_scale, probability_flow=probability_flow,
              self_conditioning=self_conditioning, device=device, rng=rng)
    rig0 = lambda bsz: Rigid.from_tensor_4x4(gt_frames_4x4.repeat(bsz, *(1,) * (gt_frames_4x4.ndim - 1)))  # noqa: E731
    out = []
    for group in merge_chunk_groups(chunks, N, mergeable=probability_flow or rng == "device", max_pairs=max_pairs):
        if len(group) == 1:
            bsz, lo, hi = group[0]
            if hi > lo or rng == "host":   # an empty slice still advances the host generators in lock-step with the other ranks
                out.append(forward_backward(net, diffuser, batch, rig0(bsz), float(t_delta), replica_slice=(lo, hi), **kw))
            continue
        starts, tail_burn = [], None
        for i, (bsz, lo, hi) in enumerate(group):
            if hi > lo or rng == "host":
                r = _start_frames(diffuser, batch, rig0(bsz), float(t_delta), lo, hi, rng, device)
                if rng == "host":
                    if i + 1 < len(group) or _skips_unused_draws(probability_flow, bsz, N):
                        _burn_step_draws(bsz, N, len(ts) - 1)   # the next chunk's start frames come after this chunk's step draws
                    else:
                        tail_burn = bsz                          # drawn for real: the last chunk's ride in the loop, behind the GPU
                if r is not None:
                    starts.append(r)
        if not starts:
            if tail_burn is not None:
                _burn_step_draws(tail_burn, N, len(ts) - 1)
            continue
        rigids_t = torch.cat(starts, dim=0) if len(starts) > 1 else starts[0]
        b = rigids_t.shape[0]
        feats = {k: batch[k].to(device).repeat(b, *(1,) * (batch[k].ndim - 1)) for k in _REPEAT_KEYS if k in batch}

        def host_noise(bsz=tail_burn):   # (a mergeable group in host mode runs the ODE: the draws are consumed, not used)
            torch.randn(bsz, N, 3, dtype=torch.float64)
            torch.randn(bsz, N, 3, dtype=torch.float64)
            return None

        out.append(denoise_loop(net, diffuser, feats, rigids_t, ts, dt, min_t=min_t, noise_scale=noise_scale,
                                probability_flow=probability_flow, self_conditioning=self_conditioning, center_mode=1,
                                host_noise=host_noise if tail_burn is not None else None)[0])
    if not out:
        return torch.zeros(0, N, 37, 3, device=device)
    return torch.cat(out, dim=0) if len(out) > 1 else out[0]


def _denoise_pass_deltas(net, diffuser, batch: dict, groups, *, min_t: float, noise_scale: float, probability_flow: bool,
                         self_conditioning: bool, center_mode: int, host_noise, device):
    """One pass over SEVERAL trajectories of one target with different schedules (the t_deltas of the reference's inference block)
    as ONE growing batch.  ``groups`` = [dict(rigids_t [b_k,N,7], ts (descending), dt)] in output order.  The trajectories are
    aligned at their END: group k (n_k steps) joins at global step n_max - n_k, after its own self-conditioning evaluation, and
    every sample carries its own timestep image (``t_img`` [b,512]), SE(3) step parameters and step size (s2s_se3_step
    dt_per_sample) -- so each sample sees exactly the evaluations of its single-t_delta run (the kernels are batch-invariant).
    -> (atom37 per group, rigids7 of the whole batch, psi per group)."""
    emb = net.embedder
    order = sorted(range(len(groups)), key=lambda k: -len(groups[k]["ts"]))     # join order (stable: equal lengths keep output order)
    n_max = len(groups[order[0]]["ts"])
    N = groups[0]["rigids_t"].shape[1]
    timg, p8s, base, o = [], [], {}, 0
    for k in order:
        t_all = torch.as_tensor(np.ascontiguousarray(groups[k]["ts"], dtype=np.float64)).float()   # fl32(t), as `t * torch.ones(B)` gives
        p8s.append(diffuser.step_params(t_all).to(device))
        timg.append(emb.time_images(emb.time_embed(t_all).to(device)))
        base[k] = o
        o += len(t_all)
    TIMG, P8 = torch.cat(timg).contiguous(), torch.cat(p8s).contiguous()
    one = {k: batch[k].to(device) for k in _REPEAT_KEYS if k in batch}

    def expand(b):
        f = {k: v.repeat(b, *(1,) * (v.ndim - 1)) for k, v in one.items()}
        for k in ("residue_mask", "fixed_mask"):     # float32 device tensors once per batch composition (the network's caches key on them)
            f[k] = f[k].float().contiguous()
        return f

    keep_bb = getattr(net, "backbone_in_forward", None)
    if keep_bb is not None:
        net.backbone_in_forward = False
    try:
        rig = sc = idx = dtv = feats = mask = diffuse_mask = None
        members, final = [], None
        for g in range(n_max):
            joined = False
            for k in order:
                if n_max - len(groups[k]["ts"]) != g:
                    continue
                rk = groups[k]["rigids_t"]
                bk = rk.shape[0]
                sck = torch.zeros(bk, N, 3, device=device)
                if self_conditioning:      # the group's extra evaluation at its first t with an empty self-conditioning input, on its own
                    fk = expand(bk)
                    fk.update(rigids_t=rk, sc_ca_t=sck, t=torch.full((bk,), float(groups[k]["ts"][0]), dtype=torch.float32),
                              t_img=TIMG[base[k]])
                    sck = _net_eval(net, fk, True)["rigids7"][..., 4:].clone()
                ik = torch.full((bk,), base[k], dtype=torch.int64, device=device)
                dk = torch.full((bk,), float(groups[k]["dt"]), dtype=torch.float64, device=device)
                rig = rk if rig is None else torch.cat([rig, rk])
                sc = sck if sc is None else torch.cat([sc, sck])
                idx = ik if idx is None else torch.cat([idx, ik])
                dtv = dk if dtv is None else torch.cat([dtv, dk])
                members.append((k, bk))
                joined = True
            b = rig.shape[0]
            if joined:
                feats = expand(b)
                mask = feats["residue_mask"]
                diffuse_mask = ((1 - feats["fixed_mask"]) * mask).contiguous()
                feats["t"] = torch.zeros(b)     # (not read: every sample's timestep enters through its t_img row)
            feats["rigids_t"], feats["sc_ca_t"] = rig, sc
            feats["t_img"] = TIMG.index_select(0, idx)
            out = _net_eval(net, feats, True)
            x0_7 = out["rigids7"]
            if g == n_max - 1:           # every trajectory's last evaluation (t == min_t): the x0 prediction is the sample
                final = out
                break
            if self_conditioning:
                sc = x0_7[..., 4:]
            z = host_noise() if host_noise is not None else None
            z_rot, z_trans = z if z is not None else (None, None)
            if not probability_flow and z_rot is None:
                z_rot = torch.randn(b, N, 3, dtype=torch.float64, device=device)
                z_trans = torch.randn(b, N, 3, dtype=torch.float64, device=device)
            rig, _, _ = diffuser.step(x0_7, rig, P8.index_select(0, idx), dtv, mask, diffuse_mask, center_trans=center_mode,
                                      noise_scale=noise_scale, probability_flow=probability_flow, z_rot=z_rot, z_trans=z_trans)
            idx = idx + 1
        if final.get("psi_deferred"):
            final = dict(final, psi=net.blend_psi(final["psi"], feats["torsion_angles_sin_cos"], feats["fixed_mask"]))
        atom37 = compute_backbone(final["rigids"], final["psi"], aatype=feats.get("aatype"), _rigids7=final["rigids7"])[0]
    finally:
        if keep_bb is not None:
            net.backbone_in_forward = keep_bb
    a_out, p_out, o = [None] * len(groups), [None] * len(groups), 0
    for k, bk in members:
        a_out[k], p_out[k] = atom37[o:o + bk], final["psi"][o:o + bk]
        o += bk
    return a_out, final["rigids7"], p_out


def merge_delta_groups(steps, b: int, N: int, max_pairs: int = None):
    """Consecutive t_deltas (``steps[i]`` = their trajectory lengths) -> groups [[i, ...]] sampled as one growing batch each: a group
    holds at most ``max_pairs`` pairs at its end (b replicas per t_delta).  ``S2S_MERGE_DELTAS=0``: one t_delta per group."""
    max_pairs = _MERGE_MAX_PAIRS if max_pairs is None else max_pairs
    per = max(1, b) * N * N
    cap = 1 if os.environ.get("S2S_MERGE_DELTAS", "1") == "0" else max(1, max_pairs // per)
    return [list(range(i, min(i + cap, len(steps)))) for i in range(0, len(steps), cap)]


def forward_backward_deltas(net, diffuser, batch: dict, gt_frames_4x4: torch.Tensor, chunks, delta_range, **kw):
    """``iter_forward_backward_deltas`` collected: [atom37 [sum(hi - lo), N, 37, 3] per t_delta] in ``delta_range`` order."""
    delta_range = list(delta_range)
    out = [None] * len(delta_range)
    for idx, a37s in iter_forward_backward_deltas(net, diffuser, batch, gt_frames_4x4, chunks, delta_range, **kw):
        for i, a in zip(idx, a37s):
            out[i] = a
    return out


def iter_forward_backward_deltas(net, diffuser, batch: dict, gt_frames_4x4: torch.Tensor, chunks, delta_range, *, num_timesteps: int,
                                 min_t: float = 0.01, noise_scale: float = 1.0, probability_flow: bool = True,
                                 self_conditioning: bool = True, device=None, rng: str = "host", max_pairs: int = None):
    """All t_deltas of one target (the outer loop of the reference's predict_step, diffusion_module.py:341-367) -> [atom37
    [sum(hi - lo), N, 37, 3] per t_delta], each exactly what ``forward_backward_chunks`` returns for that t_delta.

    The reference's default block runs 10 t_deltas (0.25 .. 0.70 of 1000 timesteps: trajectories of 250 .. 700 steps) of 100 replicas
    one after the other; on chains of 35 .. 80 residues every network evaluation is then bound by the latency of its ~83 dependent
    launches, not by the GPU.  A replica's trajectory does not depend on its batch, and nothing in the network or in the SE(3) step
    couples the samples of a batch -- so the t_deltas whose replicas fit the pair budget run as ONE batch that GROWS: aligned at
    their common end (t = min_t), the longest trajectory starts alone and each shorter one joins when as many steps remain as it
    has, with its own timestep image, step parameters and step size per sample (``_denoise_pass_deltas``).  4750 + 10 evaluations of
    100 replicas become 700 + 10 of 100 .. 1000.  The host noise stream keeps the reference's order: start frames t_delta by t_delta,
    chunk by chunk, each chunk's (unused, under the ODE) per-step draws consumed before the next chunk's start frames -- the last
    chunk's ride in the loop when its trajectory is the longest.  Under the SDE the per-step draws are part of a trajectory (host
    noise: in the reference's order; device noise: a merged batch would draw them for the growing batch, i.e. other samples than one
    t_delta at a time under the same seed): one t_delta at a time (``forward_backward_chunks``), as the reference does."""
    device = _require_hip_device(device, net)
    delta_range = [float(t) for t in delta_range]
    kw = dict(num_timesteps=num_timesteps, min_t=min_t, noise_scale=noise_scale, probability_flow=probability_flow,
              self_conditioning=self_conditioning, device=device, rng=rng)
    N = gt_frames_4x4.shape[-3]
    b_rank = sum(hi - lo for _, lo, hi in chunks)
    mergeable = probability_flow and len(delta_range) > 1 and b_rank > 0 and all(t > 0 for t in delta_range) \
        and hasattr(getattr(net, "embedder", None), "time_images")
    sched = [schedule(t, num_timesteps, min_t) for t in delta_range]
    plan = merge_delta_groups([s[1] for s in sched], b_rank, N, max_pairs) if mergeable else [[i] for i in range(len(delta_range))]
    rig0 = lambda bsz: Rigid.from_tensor_4x4(gt_frames_4x4.repeat(bsz, *(1,) * (gt_frames_4x4.ndim - 1)))  # noqa: E731
    for grp in plan:
        if len(grp) == 1:
            with torch.no_grad():   # (never yield inside the context: the caller would inherit the grad mode)
                one = forward_backward_chunks(net, diffuser, batch, gt_frames_4x4, chunks, delta_range[grp[0]], max_pairs=max_pairs, **kw)
            yield grp, [one]
            continue
        # start frames in the reference's order: t_delta by t_delta, chunk by chunk (host mode: + each chunk's step draws)
        groups, tail = [], None
        longest_last = all(sched[grp[-1]][1] >= sched[i][1] for i in grp)
        for gi, i in enumerate(grp):
            T, n, dt, ts = sched[i]
            starts = []
            for ci, (bsz, lo, hi) in enumerate(chunks):
                with torch.no_grad():
                    r = _start_frames(diffuser, batch, rig0(bsz), delta_range[i], lo, hi, rng, device)
                if rng == "host":
                    if gi + 1 == len(grp) and ci + 1 == len(chunks) and longest_last and not _skips_unused_draws(probability_flow, bsz, N):
                        tail = (bsz, len(ts) - 1)      # drawn for real: rides in the loop, behind the GPU (global step == its local step)
                    else:
                        _burn_step_draws(bsz, N, len(ts) - 1)
                if r is not None:
                    starts.append(r)
            groups.append(dict(rigids_t=torch.cat(starts) if len(starts) > 1 else starts[0], ts=ts, dt=dt))
        left = [tail[1] if tail else 0]

        def host_noise(left=left, bsz=tail[0] if tail else 0):
            if left[0] > 0:
                left[0] -= 1
                torch.randn(bsz, N, 3, dtype=torch.float64)
                torch.randn(bsz, N, 3, dtype=torch.float64)
            return None

        def run_pass(groups=groups, left=left, tail=tail, host_noise=host_noise):
            left[0] = tail[1] if tail else 0
            a, r7, _ = _denoise_pass_deltas(net, diffuser, batch, groups, min_t=min_t, noise_scale=noise_scale,
                                            probability_flow=probability_flow, self_conditioning=self_conditioning, center_mode=1,
                                            host_noise=host_noise if tail else None, device=device)
            return a, r7

        with torch.no_grad():
            a37 = _range_guarded(net, run_pass, device, host_draws=tail is not None, device_draws=False)[0]
        yield grp, list(a37)


def forward_flops(n_res: int) -> float:
    """Algorithmic FLOPs of one network evaluation per replica: F(N) = 2,251,264 N^2 + 3.24e7 N."""
    return 2251264.0 * n_res * n_res + 3.24e7 * n_res


def plan_mixed_work(lengths, replicas: int, world: int = 1, *, max_pairs: int = 24 << 20, ms_per_mpair: float = 10.7,
                    launch_floor_ms: float = 4.0):
    """Work distribution for many chains of different length (BASELINE configs[4]): flatten to (chain,
    replica block) items, balance them over the ranks by FLOP weight, bucket each rank's items by padded length.

      1. every chain's replicas are ceil-split into ``world`` blocks (``shard_range``); the non-empty blocks are the items,
         weight = (replicas in the block) x F(N);
      2. longest-processing-time-first: items in decreasing weight go to the currently least loaded rank (even splits give
         every rank the same set of lengths; remainders and replicas < world are what the weights are for);
      3. per rank, items in decreasing length are packed into padded batches (n_pad = the longest chain of the batch; the
         kernels take ragged N, so there is no tile rounding).  An item joins the open batch when that is cheaper than a
         batch of its own (within 5 %) under  cost(batch) = max(launch_floor_ms, ms_per_mpair x padded Mpairs)  per network evaluation
         -- one evaluation is ~83 dependent launches, so small batches are launch-bound and padding them into a neighbour is free,
         while large ones pay for every padded pair (measured: 10.7 ms per 2^20 pairs at cfg2) -- and the batch stays
         below ``max_pairs`` (device memory: ~1.3 KB per pair).
    -> plan[rank] = [ {"n_pad": int, "items": [(chain, replica_lo, replica_hi), ...]}, ... ]"""
    lengths = [int(x) for x in lengths]
    items = []
    for k, L in enumerate(lengths):
        for r in range(world):
            lo, hi = shard_range(replicas, r, world)
            if hi > lo:
                items.append((forward_flops(L) * (hi - lo), k, lo, hi))
    items.sort(key=lambda it: (-it[0], it[1], it[2]))
    load = [0.0] * world
    mine = [[] for _ in range(world)]
    for w, k, lo, hi in items:
        r = min(range(world), key=lambda q: (load[q], q))
        load[r] += w
        mine[r].append((k, lo, hi))
    plan = []
    for r in range(world):
        cost = lambda pairs: max(launch_floor_ms, ms_per_mpair * pairs / float(1 << 20))  # noqa: E731
        batches, cur, cur_b, n_pad = [], [], 0, 0
        for k, lo, hi in sorted(mine[r], key=lambda it: (-lengths[it[0]], it[0], it[1])):
            L, b = lengths[k], hi - lo
            if cur:
                merged = (cur_b + b) * n_pad * n_pad
                if merged > max_pairs or cost(merged) > 1.05 * (cost(cur_b * n_pad * n_pad) + cost(b * L * L)):
                    batches.append({"n_pad": n_pad, "items": cur})
                    cur, cur_b = [], 0
            if not cur:
                n_pad = L
            cur.append((k, lo, hi)); cur_b += b
        if cur:
            batches.append({"n_pad": n_pad, "items": cur})
        plan.append(batches)
    return plan


def mixed_batch_seed(base_seed: int, t_delta: float, chain: int, replica_lo: int) -> int:
    """Seed of the host generators for the padded batch whose first item is (chain, replica_lo): a function of the run seed and the
    work item only, so no two batches of a run -- on one rank or on different ranks -- share a noise stream."""
    return (int(base_seed) * 1000003 + int(round(float(t_delta) * 1000)) * 998244353 + int(chain) * 7919 + int(replica_lo) * 104729
            + 12345) % (2 ** 63 - 1)


@torch.no_grad()
def sample_mixed_lengths(net, diffuser, targets, replicas: int, t_delta: float, *, num_timesteps: int,
                         min_t: float = 0.01, noise_scale: float = 1.0, probability_flow: bool = True,
                         self_conditioning: bool = True, device=None, rigids_t_init=None, shard: Tuple[int, int] = (0, 1),
                         rng: str = "host", max_pairs: int = 24 << 20, plan=None, seed_base: Optional[int] = None):
    """BASELINE configs[4]: many chains of different length, ``replicas`` each, in padded batches.

    The reference cannot do this (`assert batch size == 1`, diffusion_module.py:249) and its padding semantics
    would be wrong for it (float key-padding mask added to the transformer logits, centre of mass over padded
    residues).  Here padding is exact: padded keys are removed from both attentions, pair rows/cols are
    masked, the centre of mass runs over real residues only -- so every chain reproduces its own un-padded run.
    Work is distributed by ``plan_mixed_work`` (FLOP-weighted over ``shard`` = (rank, world), bucketed by length).
    ``targets``: list of single-target feature dicts (batch dim 1).  ``rng="device"`` draws the forward-marginal / step
    noise on the device; "host" on the host generator per (chain, block) in plan order; ``rigids_t_init[k]`` [replicas,
    N_k, 7] overrides the starting frames.
    -> list over chains of (replica_lo, atom37 [b, N_k, 37, 3]) pieces this rank sampled, in replica order."""
    device = torch.device(device) if device is not None else next(net.parameters()).device
    lengths = [int(t["aatype"].shape[1]) for t in targets]
    rank, world = shard
    if plan is None:
        plan = plan_mixed_work(lengths, replicas, world, max_pairs=max_pairs)
    T, n, dt, ts = schedule(t_delta, num_timesteps, min_t)
    pieces = [[] for _ in targets]
    tr = net.translator
    keep = tr.exact_padding
    tr.exact_padding = True
    base_seed = int(torch.initial_seed()) if seed_base is None else int(seed_base)
    # The per-batch seeding below re-seeds the process-global host generators; their states are put back on the way out, so that the
    # call leaves no trace in them: `torch.initial_seed()` stays the run seed for the next call (a loop over t_delta without
    # ``seed_base`` sees the same base every time) and whatever the caller samples afterwards continues ITS stream.
    host_states = (torch.get_rng_state(), np.random.get_state()) if rng == "host" else None
    try:
        for batch in plan[rank]:
            n_pad = batch["n_pad"]
            if rng == "host":
                # The host generators are seeded per batch from (run seed, t_delta, first (chain, replica) item of the batch): every
                # rank starts from the same run seed (eval.py), and without this each rank would draw the noise of ITS blocks from the
                # same stream -- identical starting frames and step noise for replica blocks 0-24, 25-49, ... of a chain, i.e. duplicate
                # conformations in the gathered ensemble.  (The device mode offsets its Philox seed by the rank instead.)
                ti0, lo0, _ = batch["items"][0]
                item_seed = mixed_batch_seed(base_seed, t_delta, ti0, lo0)
                torch.random.default_generator.manual_seed(item_seed)   # the CPU generator only: the device stream is not touched
                np.random.seed(item_seed % (2 ** 32))
            rows, r_t = {k: [] for k in _REPEAT_KEYS}, []
            for ti, lo, hi in batch["items"]:
                tg, L, b = targets[ti], lengths[ti], hi - lo
                for k in _REPEAT_KEYS:
                    v = tg[k]
                    pad = torch.zeros((1, n_pad - L) + tuple(v.shape[2:]), dtype=v.dtype)
                    if k == "residue_idx" and L < n_pad:  # keep padded indices inside the real range (table lookups stay in bounds)
                        pad = pad + v[:, -1:]
                    rows[k].append(torch.cat([v.cpu(), pad], dim=1).repeat(b, *(1,) * (v.ndim - 1)))
                gt4 = tg["rigidgroups_gt_frames"][..., 0, :, :]
                if rigids_t_init is not None:
                    rt = rigids_t_init[ti][lo:hi].to(device).float()
                elif rng == "device":
                    if t_delta > 0:
                        rt = diffuser.forward_marginal_device(gt4.to(device).float().repeat(b, 1, 1, 1), t_delta,
                                                              tg["residue_mask"].to(device).float().repeat(b, 1))
                    else:
                        rt = diffuser.forward_marginal_device(None, None, shape=(b, L))
                else:
                    rig0 = Rigid.from_tensor_4x4(gt4.cpu().repeat(b, 1, 1, 1))
                    if t_delta > 0:
                        rt = diffuser.forward_marginal(rig0, t_delta * torch.ones(b), tg["residue_mask"].cpu().repeat(b, 1))["rigids_t"]
                    else:
                        rt = diffuser.sample_prior(shape=rig0.shape, device="cpu", as_tensor_7=True)["rigids_t"]
                    rt = rt.to(device).float()
                ident = torch.zeros(b, n_pad - L, 7, device=device)
                ident[..., 0] = 1.0  # identity frames on the padding: finite everywhere, masked out of every result
                r_t.append(torch.cat([rt, ident], dim=1))
            feats = {k: torch.cat(v, dim=0).to(device) for k, v in rows.items()}
            rigids_t = torch.cat(r_t, dim=0).contiguous()
            atom37, _, _ = denoise_loop(net, diffuser, feats, rigids_t, ts, dt, min_t=min_t, noise_scale=noise_scale,
                                        probability_flow=probability_flow, self_conditioning=self_conditioning, center_mode=2)
            o = 0
            for ti, lo, hi in batch["items"]:
                pieces[ti].append((lo, atom37[o:o + hi - lo, :lengths[ti]].clone()))
                o += hi - lo
    finally:
        tr.exact_padding = keep
        if host_states is not None:
            torch.set_rng_state(host_states[0])
            np.random.set_state(host_states[1])
    return [sorted(p, key=lambda x: x[0]) for p in pieces]
