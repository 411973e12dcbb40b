"""Folded against per-head projections of one InvariantPointAttention block (models/net/ipa.py: _folded_packs): time of
projections -> points -> attention -> linear_out and the difference of the block's output, each against the exact fp32 path.
    python tools/ipa_fold_ab.py [--B 128 --N 256 --sigma 0.05]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ap = argparse.ArgumentParser()
ap.add_argument("--B", type=int, default=128)
ap.add_argument("--N", type=int, default=256)
ap.add_argument("--iters", type=int, default=5)
ap.add_argument("--sigma", type=float, default=0.05)
ap.add_argument("--only-folded", action="store_true", help="run the folded (production) form alone: what a PMC pass should see")
a = ap.parse_args()
from str2str_amd import ops  # noqa: E402
from str2str_amd.arith import use_arith  # noqa: E402
from str2str_amd.models.net.ipa import InvariantPointAttention  # noqa: E402

torch.manual_seed(0)
B, N, H = a.B, a.N, 8
M = B * N
ipa = InvariantPointAttention(256, 128, 256, 8, 8, 12).cuda()
with torch.no_grad():
    for p in ipa.parameters():
        p.copy_(torch.randn_like(p) * a.sigma)
g = torch.Generator(device="cuda").manual_seed(0)
rn = lambda *s: torch.randn(*s, device="cuda", generator=g)  # noqa: E731
s = rn(M, 256)
quat = rn(B, N, 4)
r7 = torch.cat([quat / quat.norm(dim=-1, keepdim=True), rn(B, N, 3)], -1).contiguous()
bias, pz = rn(B, H, N, N), rn(B, N, N, 32)
mask = torch.ones(B, N, device="cuda")
s_xp = ops.pack_planes(s)


def block(act):
    feats = ipa.attention(act, B, N, r7, mask, (bias.clone(), pz))
    return ops.node_apply(feats, ipa.out_pack(feats), M)[0]


def timed(name, fn):
    for _ in range(2):
        out = fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.iters):
        out = fn()
    e1.record()
    torch.cuda.synchronize()
    print(f"  {name:28s} {e0.elapsed_time(e1) / a.iters:8.3f} ms", flush=True)
    return out


with torch.no_grad():
    print(f"B={B} N={N} sigma={a.sigma}")
    if a.only_folded:
        ipa.arith, ipa.fold = "f16x3", True
        timed("folded (K = V = s)", lambda: block(s_xp))
        sys.exit(0)
    ipa.arith = "f32"
    ref = timed("exact fp32 path", lambda: block(s)).double()
    ipa.arith = "f16x3"
    for fold in (False, True, False, True):
        ipa.fold = fold
        out = timed("folded (K = V = s)" if fold else "per-head k / v projections", lambda: block(s_xp)).double()
        print(f"    vs fp32 path: max |diff| / max |ref| = {((out - ref).abs().max() / ref.abs().max()).item():.2e}")
