"""Run only the IPA attention core (s2s_ipa_attention + s2s_ipa_opair) for PMC passes / quick timing."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ap = argparse.ArgumentParser()
ap.add_argument("--B", type=int, default=128)
ap.add_argument("--N", type=int, default=256)
ap.add_argument("--iters", type=int, default=5)
a = ap.parse_args()
from str2str_amd import ops  # noqa: E402

B, N, H, C = a.B, a.N, 8, 256
g = torch.Generator(device="cuda").manual_seed(0)
rn = lambda *s: torch.randn(*s, device="cuda", generator=g)
q, kv = rn(B, N, H, C), rn(B, N, H, 2 * C)
quat = rn(B, N, 4)
r7 = torch.cat([quat / quat.norm(dim=-1, keepdim=True), rn(B, N, 3)], -1).contiguous()
qp, kp = rn(B, N, H, 24), rn(B, N, H, 24)
vp = torch.zeros(B, N, H, 64, device="cuda")
vp.view(B, N, H, 16, 4)[..., :12, :3] = rn(B, N, H, 12, 3)
bias, pz = rn(B, H, N, N), rn(B, N, N, 32)
mask = torch.ones(B, N, device="cuda")
hw = torch.full((H,), 0.1, device="cuda")
for _ in range(2):
    ops.ipa_attention(q, kv, qp, kp, vp, bias, pz, mask, r7, hw)
torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(a.iters):
    ops.ipa_attention(q, kv, qp, kp, vp, bias, pz, mask, r7, hw)
e.record()
torch.cuda.synchronize()
ms = s.elapsed_time(e) / a.iters
print(f"B={B} N={N}: {ms:.3f} ms per (attention + o_pair); algorithmic {B * 4 * (9512 * N + 40 * N * N) / ms / 1e6:.0f} GB/s")
