"""s2s_node_chain against the separate launches: NodeTransition (3 x 256, residual + LayerNorm) and an encoder feed-forward pair
(2 x 320, residual + LayerNorm) at several row counts.   python tools/node_chain_bench.py [--iters 50]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ap = argparse.ArgumentParser()
ap.add_argument("--iters", type=int, default=50)
a = ap.parse_args()
from str2str_amd import ops  # noqa: E402

dev = "cuda"
g = torch.Generator(device=dev).manual_seed(0)
rn = lambda *s: torch.randn(*s, device=dev, generator=g)  # noqa: E731


def timeit(fn):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(a.iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / a.iters * 1e3


for width, n in ((256, 3), (320, 2)):
    layers = [ops.pack_node_layer(rn(width, width) / width ** 0.5, rn(width), i == n - 1) for i in range(n)]
    gam, bet = rn(width), rn(width)
    relu = tuple(i < n - 1 for i in range(n))
    for M in (1260, 2240, 5120, 32768, 80000):
        x, res = ops.pack_planes(rn(M, width)), rn(M, width)
        kw = dict(residual=res, ln=(gam, bet, 1e-5), want_xp=True)

        def separate():
            act = x
            for i, L in enumerate(layers[:-1]):
                _, act = ops.node_apply(act, L, M, relu=relu[i], want_f32=False, want_xp=True)
            return ops.node_apply(act, layers[-1], M, relu=False, **kw)

        chain = lambda: ops.node_apply_chain(x, layers, M, relu, **kw)  # noqa: E731
        w, c = separate(), chain()
        same = torch.equal(w[0], c[0]) and torch.equal(w[1], c[1])
        print(f"{n} x {width}  M = {M:6d}: separate {timeit(separate):7.1f} us   chain {timeit(chain):7.1f} us   bitwise equal: {same}", flush=True)
