"""The 16-pair edge embedding (csrc/edge_embed16.hip, S2S_EE_KERNEL=16) against the 32-pair kernel: outputs (pair tensor, attention bias,
pair_z) on the same inputs, both layouts, ragged sizes; then HIP-event times of both at the cfg2 shape.
    python tools/ee16_check.py [--time]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from str2str_amd import ops  # noqa: E402
from str2str_amd.factory import build_synthetic_net  # noqa: E402

net = build_synthetic_net(device="cuda")
emb = net.embedder
proj = net.translator.trunk["ipa_0"].pair_proj_weights()


def run(B, N, layout, k16, with_proj=True, mask_p=0.1, seed=0):
    g = torch.Generator(device="cuda").manual_seed(seed)
    idx = (torch.arange(N)[None] + torch.randint(0, 5, (B, 1))).repeat(1, 1) if False else torch.arange(N)[None].repeat(B, 1)
    ca = torch.randn(B, N, 3, device="cuda", generator=g) * 6
    fixed = torch.zeros(B, N, device="cuda")
    mask = (torch.rand(B, N, device="cuda", generator=g) > mask_p).float()
    t_emb = emb.time_embed(torch.full((1,), 0.37)).to("cuda")
    emb.kernel16 = k16
    with torch.no_grad():
        r = emb(idx, None, fixed, ca, node_mask=mask, next_proj=proj if with_proj else None, t_emb=t_emb, edge_layout=layout)
    z = r[1]
    z = ops.pair_untiled(z) if isinstance(z, ops.PairTiled) else z
    return (z,) + (tuple(r[2]) if with_proj else ())


worst = 0.0
for B, N in [(2, 32), (3, 47), (1, 10), (5, 64), (2, 100), (16, 256)]:
    for layout in ("rowmajor", "tiled"):
        for wp in (True, False):
            if layout == "tiled" and not wp:
                continue
            a = run(B, N, layout, False, wp)
            b = run(B, N, layout, True, wp)
            for name, x, y in zip(("z", "bias", "pair_z"), a, b):
                err = float((x - y).abs().max() / x.abs().max().clamp_min(1e-30)) if torch.isfinite(y).all() else float("inf")
                worst = max(worst, err)
                if err >= 5e-6:
                    print("MISMATCH", (B, N, layout, wp, name, err), flush=True)
print("ee16 vs ee32: worst relative difference", worst, "(OK)" if worst < 5e-6 else "(FAILED)")
if "--time" in sys.argv:
    for k16 in (False, True, False, True):
        emb.kernel16 = k16
        B, N = 128, 256
        idx = torch.arange(N)[None].repeat(B, 1)
        ca = torch.randn(B, N, 3, device="cuda") * 10
        fixed = torch.zeros(B, N, device="cuda"); mask = torch.ones(B, N, device="cuda")
        t_emb = emb.time_embed(torch.full((1,), 0.5)).to("cuda")
        f = lambda: emb(idx, None, fixed, ca, node_mask=mask, next_proj=proj, t_emb=t_emb, edge_layout="tiled")
        with torch.no_grad():
            for _ in range(3):
                f()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                f()
            e1.record(); torch.cuda.synchronize()
        print("kernel16" if k16 else "kernel32", "embedder ms:", round(e0.elapsed_time(e1) / 20, 3), flush=True)
