"""Run only the EdgeTransition kernel (for PMC passes / quick timing).
    python tools/et_only.py --B 16 --N 256 --mode f16x3|f32 [--proj]      (--proj: with the fused next-block pair projection)"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

ap = argparse.ArgumentParser()
ap.add_argument("--B", type=int, default=16)
ap.add_argument("--N", type=int, default=256)
ap.add_argument("--iters", type=int, default=3)
ap.add_argument("--mode", default="f16x3", choices=["f16x3", "f32"])
ap.add_argument("--proj", action="store_true")
ap.add_argument("--layout", default="rowmajor", choices=["rowmajor", "tiled", "none"], help="f16x3: pair-tensor layout (in and out)")
a = ap.parse_args()
os.environ["S2S_ARITH"] = a.mode
from str2str_amd.factory import build_synthetic_net  # noqa: E402

net = build_synthetic_net(device="cuda")
et = net.translator.trunk["edge_transition_0"]
kw = {"next_proj": net.translator.trunk["ipa_1"].pair_proj_weights()} if a.proj else {}
g = torch.Generator(device="cuda").manual_seed(0)
node = torch.randn(a.B, a.N, 256, device="cuda", generator=g)
edge = torch.randn(a.B, a.N, a.N, 128, device="cuda", generator=g)
mask = torch.ones(a.B, a.N, device="cuda")
if a.layout != "rowmajor":   # the trunk's chaining: tiled in, tiled (or no) out; per-node parts outside the timed call
    from str2str_amd import ops
    n_p, node_ab = et.node_parts(ops.to_act(node.reshape(a.B * a.N, -1).contiguous(), "f16x3"), a.B * a.N, kernel_form=True)
    zt = ops.pair_tiled(edge)
    del edge
    et = lambda *_, **__: net.translator.trunk["edge_transition_0"].pair_mlp(zt, node_ab.view(a.B, a.N, -1), n_p.view(a.B, a.N, -1), mask,
                                                                          kw.get("next_proj"), out_layout=a.layout, ab_kernel_form=True)
    edge = None
with torch.no_grad():
    for _ in range(2):
        et(node, edge, edge_mask_1d=mask, **kw)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(a.iters):
        et(node, edge, edge_mask_1d=mask, **kw)
    e.record()
    torch.cuda.synchronize()
ms = s.elapsed_time(e) / a.iters
pairs = a.B * a.N * a.N
print(f"mode={a.mode} proj={a.proj} layout={a.layout} B={a.B} N={a.N}: {ms:.3f} ms/launch  fp32-equivalent {pairs * 491520 / ms / 1e9:.1f} TFLOP/s")
