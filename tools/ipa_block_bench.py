"""Time the node-side of one InvariantPointAttention block (projections -> points -> attention core -> packed linear_out input)
stage by stage, on the pre-split f16 operand path (any length) and on the fp32-operand path (arith "f32")."""
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
from str2str_amd.models.net.ipa import InvariantPointAttention  # noqa: E402

torch.manual_seed(0)
B, N, H, C = a.B, a.N, 8, 256
M = B * N
ipa = InvariantPointAttention(256, 128, 256, 8, 8, 12).cuda()
with torch.no_grad():
    for p in ipa.parameters():
        p.copy_(torch.randn_like(p) * 0.05)
g = torch.Generator(device="cuda").manual_seed(0)
rn = lambda *s: torch.randn(*s, device="cuda", generator=g)
s = rn(M, 256)
quat = rn(B, N, 4)
r7 = torch.cat([quat / quat.norm(dim=-1, keepdim=True), rn(B, N, 3)], -1).contiguous()
bias, pz = rn(B, H, N, N), rn(B, N, N, 32)
mask = torch.ones(B, N, device="cuda")
s_xp = ops.pack_planes(s)
w, d = ipa.node_packs(), ipa._derived()


def timeit(name, fn):
    for _ in range(2):
        out = fn()
    torch.cuda.synchronize()
    s_, e_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s_.record()
    for _ in range(a.iters):
        out = fn()
    e_.record()
    torch.cuda.synchronize()
    ms = s_.elapsed_time(e_) / a.iters
    print(f"  {name:34s} {ms:8.3f} ms", flush=True)
    return out, ms


lin = lambda x, **kw: ops.node_apply(s_xp, x, M, **kw)  # noqa: E731
NP = ops.padded_len(N)
rmap, Mo = ((NP, N), B * NP) if NP != N else (None, M)
linp = lambda x, **kw: ops.node_apply(s_xp, x, Mo, row_map=rmap, **kw)  # noqa: E731
alg = B * 4 * (9512 * N + 40 * N * N)
with torch.no_grad():
    if True:
        print(f"f16 pair operand path (s2s_ipa_attention_f16w)  B={B} N={N} (padded {NP})")
        tot = 0.0
        (_, q_xp), t = timeit("q  -> planes", lambda: linp(w["q"], want_f32=False, want_xp=True)); tot += t
        (_, k_xp), t = timeit("k  -> planes", lambda: linp(w["k"], want_f32=False, want_xp=True)); tot += t
        v_vf, t = timeit("v  -> A fragments", lambda: ops.node_linear_vfrag(s_xp, w["v"]["w"], w["v"]["b"], Mo, 256, 2048, 8, row_map=rmap)); tot += t
        (qp, _), t = timeit("q points (linear)", lambda: lin(w["qp"])); tot += t
        (kvp, _), t = timeit("kv points (linear)", lambda: lin(w["kvp"])); tot += t
        pts, t = timeit("points -> fragments", lambda: ops.ipa_prep_points_f16(r7, qp, kvp, d["hw"])); tot += t
        (feats, fxp), t_att = timeit("attention + o_pair", lambda: ops.ipa_attention_f16(q_xp, k_xp, v_vf, pts, bias, pz, mask, r7)); tot += t_att
        f2 = feats.view(M, -1)
        _, t = timeit("pack o_pt | o_pair", lambda: ops.pack_planes(f2, col0=2048, n_cols=640, out=fxp, out_k=2688, k0=2048)); tot += t
        print(f"  total {tot:.3f} ms; attention + o_pair: algorithmic {alg / t_att / 1e6:.0f} GB/s = {alg / t_att / 1e6 / 80:.1f} % of 8 TB/s")
    print("fp32-operand path")
    tot = 0.0
    (q, _), t = timeit("q", lambda: lin(w["q"])); tot += t
    (kv, _), t = timeit("kv", lambda: lin(w["kv"])); tot += t
    _, t = timeit("q points (linear)", lambda: lin(w["qp"])); tot += t
    _, t = timeit("kv points (linear)", lambda: lin(w["kvp"])); tot += t
    (q_pts, k_pts, v_pts), t = timeit("points", lambda: ops.ipa_prep_points(r7, qp.view(B, N, -1), kvp.view(B, N, -1), 8, 8, 12)); tot += t
    feats0, t_att = timeit("attention + o_pair", lambda: ops.ipa_attention(q.view(B, N, H, -1), kv.view(B, N, H, -1), q_pts, k_pts, v_pts, bias, pz, mask, r7, d["hw"])); tot += t_att
    _, t = timeit("pack features", lambda: ops.pack_planes(feats0.view(M, -1))); tot += t
    print(f"  total {tot:.3f} ms; attention + o_pair: algorithmic {alg / t_att / 1e6:.0f} GB/s = {alg / t_att / 1e6 / 80:.1f} % of 8 TB/s")
    got = ops.unpack_planes(fxp, M, 2688)
    ref = feats0.view(M, -1)
    for name, sl in (("o", slice(0, 2048)), ("o_pt", slice(2048, 2432)), ("o_pair", slice(2432, 2688))):
        err = (got[:, sl] - ref[:, sl]).abs().max().item() / ref[:, sl].abs().max().item()
        print(f"  planes vs fp32-operand {name}: max abs err / max |ref| = {err:.2e}")
