"""Where does a workgroup of the node GEMM spend its time?  Needs the probe build:
    UNIT=node_gemm bash tools/build_variant.sh ngprobe -DS2S_NODE_PROBE
    STR2STR_HIP_LIB=str2str_amd/csrc/build/ab_ngprobe.so python tools/node_gemm_probe.py
s_memtime around the prologue, the k loop (compute / weight copy / barrier of the even k-steps) and the epilogue, wave 0 of every
workgroup (two workgroups share a CU: a wave's wall time includes what its SIMD neighbour executes meanwhile)."""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from str2str_amd import ops  # noqa: E402

lib = ops.load_library()
lib.s2s_node_probe_read.argtypes = [ctypes.c_void_p, ctypes.c_int]
M, dev = 32768, "cuda"
buf = np.zeros(8, dtype=np.uint64)
for K, N, whole, planes in [(256, 2048, False, True), (256, 2048, False, False), (2688, 256, True, True), (320, 960, False, True), (256, 256, True, True)]:
    x = torch.randn(M, K, device=dev)
    w = torch.randn(N, K, device=dev) / K ** 0.5
    b = torch.randn(N, device=dev)
    tg = ops.node_tiles(N, whole_row=whole)
    wpk, xp = ops.pack_node_weight(w, tg), ops.pack_planes(x)
    out, oxp = torch.empty(M, N, device=dev), (ops.xp_alloc(M, N, dev) if planes else None)
    run = lambda: ops.node_linear(xp, wpk, b, M, K, N, tg, out_f32=out, out_xp=oxp)
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    lib.s2s_node_probe_read(buf.ctypes.data, 1)
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(5):
        run()
    e.record()
    torch.cuda.synchronize()
    lib.s2s_node_probe_read(buf.ctypes.data, 1)
    n, ks = float(buf[6]), float(buf[7]) / float(buf[6])
    pro, cmp_, cp, bar, loop, epi = (float(buf[i]) / n for i in range(6))
    mf = ks * 3 * tg * 32
    print(f"K={K} N={N} TG={tg} planes={planes}: {s.elapsed_time(e) / 5 * 1e3:.1f} us/launch; per workgroup (ticks): prologue {pro:.0f}, k loop {loop:.0f} "
          f"({ks:.0f} k-steps; this wave's MFMAs alone {mf:.0f}), epilogue {epi:.0f}; per even k-step: compute {2 * cmp_ / ks:.0f}, copy issue {2 * cp / ks:.0f}, "
          f"barrier {2 * bar / ks:.0f} (MFMAs of one k-step: {3 * tg * 32})")
