"""Time s2s_node_linear against torch (rocBLAS fp32) on the trunk's per-node layer shapes.   python tools/node_gemm_bench.py [--M 32768]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from str2str_amd import ops  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--M", type=int, default=32768)
ap.add_argument("--iters", type=int, default=20)
ap.add_argument("--tg", type=int, default=0, help="force tiles per column block where the shape allows it (0: ops.node_tiles)")
a = ap.parse_args()
M = a.M
dev = "cuda"


def timeit(fn, iters):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters


for K, N, whole in [(256, 4096, False), (256, 2048, False), (256, 480 + 32, False), (2688, 256, True), (320, 960, False), (320, 320, True),
                    (256, 256, True), (256, 128, False), (128, 768, False)]:
    x = torch.randn(M, K, device=dev)
    w = torch.randn(N, K, device=dev) / K ** 0.5
    b = torch.randn(N, device=dev)
    tg = ops.node_tiles(N, whole_row=whole)
    if a.tg and not whole and (N // 32) % a.tg == 0:
        tg = a.tg
    wpk = ops.pack_node_weight(w, tg)
    xp = ops.pack_planes(x)
    out = torch.empty(M, N, device=dev)
    oxp = ops.xp_alloc(M, N, dev)
    t_own = timeit(lambda: ops.node_linear(xp, wpk, b, M, K, N, tg, out_f32=out, out_xp=oxp), a.iters)
    t_f32only = timeit(lambda: ops.node_linear(xp, wpk, b, M, K, N, tg, out_f32=out), a.iters)
    t_xponly = timeit(lambda: ops.node_linear(xp, wpk, b, M, K, N, tg, out_xp=oxp), a.iters)
    t_blas = timeit(lambda: torch.nn.functional.linear(x, w, b), a.iters)
    t_pack = timeit(lambda: ops.pack_planes(x, out=xp), a.iters)
    fl = 2.0 * M * K * N
    print(f"K={K:5d} N={N:5d} TG={tg:2d}: own {t_own*1e3:7.1f} us ({fl/t_own/1e9:6.1f} TF-eq)  f32-out only {t_f32only*1e3:7.1f} us  planes only {t_xponly*1e3:7.1f} us  "
          f"rocBLAS {t_blas*1e3:7.1f} us ({fl/t_blas/1e9:6.1f} TF)  pack_planes(x) {t_pack*1e3:6.1f} us", flush=True)
