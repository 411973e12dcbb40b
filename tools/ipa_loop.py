"""Loop only the f16 attention launch pair (for power / clock sampling).  python tools/ipa_loop.py [--seconds 12]"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ap = argparse.ArgumentParser()
ap.add_argument("--seconds", type=float, default=12)
ap.add_argument("--per-head", action="store_true", help="per-head k / v operands instead of the shared ones of the folded projections")
a = ap.parse_args()
from str2str_amd import ops  # noqa: E402
from str2str_amd.models.net.ipa import InvariantPointAttention  # noqa: E402

torch.manual_seed(0)
B, N, H = 128, 256, 8
M = B * N
ipa = InvariantPointAttention(256, 128, 256, 8, 8, 12).cuda()
g = torch.Generator(device="cuda").manual_seed(0)
rn = lambda *s: torch.randn(*s, device="cuda", generator=g)
s = rn(M, 256)
quat = rn(B, N, 4)
r7 = torch.cat([quat / quat.norm(dim=-1, keepdim=True), rn(B, N, 3)], -1).contiguous()
bias, pz = rn(B, H, N, N), rn(B, N, N, 32)
mask = torch.ones(B, N, device="cuda")
s_xp = ops.pack_planes(s)
w, d = ipa.node_packs(), ipa._derived()
lin = lambda x, **kw: ops.node_apply(s_xp, x, M, **kw)  # noqa: E731
with torch.no_grad():
    qp, _ = lin(w["qp"]); kvp, _ = lin(w["kvp"])
    if a.per_head:
        _, q_xp = lin(w["q"], want_f32=False, want_xp=True)
        _, k_xp = lin(w["k"], want_f32=False, want_xp=True)
        v_vf = ops.node_linear_vfrag(s_xp, w["v"]["w"], w["v"]["b"], M, 256, 2048, 8)
        pts = ops.ipa_prep_points_f16(r7, qp, kvp, d["hw"])
    else:   # the default path of the trunk: K = V = s (models/net/ipa.py _folded_packs)
        _, q_xp = lin(w["qf"], want_f32=False, want_xp=True)
        *pts, k_sh, v_vf = ops.ipa_prep_points_f16(r7, qp, kvp, d["hw"], s_xp=s_xp)
        k_xp = s_xp if k_sh is None else k_sh
    t0 = time.time(); n = 0
    while time.time() - t0 < a.seconds:
        for _ in range(50):
            ops.ipa_attention_f16(q_xp, k_xp, v_vf, pts, bias, pz, mask, r7)
        torch.cuda.synchronize(); n += 50
    print(f"{n} launch pairs in {time.time() - t0:.1f} s -> {(time.time() - t0) / n * 1e3:.3f} ms per pair")
