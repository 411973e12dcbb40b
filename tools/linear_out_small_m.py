"""linear_out (K = 2688 -> 256, + mask, residual, LayerNorm) at small row counts: fused launch vs narrow-block GEMM + s2s_row_layernorm.
    python tools/linear_out_small_m.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from str2str_amd import ops  # noqa: E402


def timeit(fn, iters=50):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3


dev = "cuda"
w = torch.randn(256, 2688, device=dev) / 50
lo = ops.pack_node_layer(w, torch.zeros(256, device=dev), True)
g, b = torch.ones(256, device=dev), torch.zeros(256, device=dev)
for M in ((1260, 2240, 2880, 5120, 8192) if len(sys.argv) < 2 else [int(a) for a in sys.argv[1:]]):
    fx = ops.pack_planes(torch.randn(M, 2688, device=dev))
    res, pm = torch.randn(M, 256, device=dev), torch.ones(M, device=dev)
    kw = dict(pre_mask=pm, residual=res, ln=(g, b, 1e-5), post_mask=pm, want_xp=True)
    fused = timeit(lambda: ops.node_linear(fx, lo["w"], lo["b"], M, 2688, 256, 8, **kw))
    split = timeit(lambda: ops.node_apply(fx, lo, M, **kw))
    pre = torch.empty(M, 256, device=dev)
    parts = []
    for tg in (1, 2, 4):
        wn = ops.pack_node_weight(w, tg)
        parts.append(timeit(lambda: ops.node_linear(fx, wn, lo["b"], M, 2688, 256, tg, pre_mask=pm, residual=res, out_f32=pre)))
    ln = timeit(lambda: ops.row_layernorm(pre, M, 256, g, b, 1e-5, post_mask=pm, want_xp=True))
    print(f"M={M:5d}: fused {fused:6.1f} us   split {split:6.1f} us   (GEMM alone tg=1/2/4: {parts[0]:.1f} / {parts[1]:.1f} / {parts[2]:.1f}, LayerNorm alone {ln:.1f})")
