import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from str2str_amd import ops
from str2str_amd.factory import build_synthetic_net
net = build_synthetic_net(device="cuda")
ipa = net.translator.trunk["ipa_1"]
DEV = "cuda"
B, N, H = 3, 64, 8
M = B * N
g = torch.Generator().manual_seed(11)
s = torch.randn(M, 256, generator=g).to(DEV)
q4 = torch.randn(B, N, 4, generator=g)
r7 = torch.cat([q4 / q4.norm(dim=-1, keepdim=True), torch.randn(B, N, 3, generator=g)], -1).contiguous().to(DEV)
bias = torch.randn(B, H, N, N, generator=g).to(DEV)
pz = torch.randn(B, N, N, 32, generator=g).to(DEV)
mask = torch.ones(B, N).to(DEV)
with torch.no_grad():
    w, d = ipa.node_packs(), ipa._derived()
    s_xp = ops.pack_planes(s)
    lin = lambda x, **kw: ops.node_linear(s_xp, x["w"], x["b"], M, x["k"], x["n"], x["tg"], **kw)
    qf, q_xp = lin(w["q"], want_f32=True, want_xp=True)
    print("q planes decode err", (ops.unpack_planes(q_xp, M, 2048) - qf).abs().max().item(), qf.abs().max().item())
    sd = ops.unpack_planes(s_xp, M, 256)
    print("s planes decode err", (sd - s).abs().max().item())
    qref = s.double() @ ipa.linear_q.weight.double().t() + ipa.linear_q.bias.double()
    print("q vs float64", (qf.double() - qref).abs().max().item() / qref.abs().max().item())
    _, k_xp = lin(w["k"], want_f32=False, want_xp=True)
    v_vf = ops.node_linear_vfrag(s_xp, w["v"]["w"], w["v"]["b"], M, 256, 2048, 8, f16=True)
    qp, _ = lin(w["qp"]); kvp, _ = lin(w["kvp"])
    pts = ops.ipa_prep_points_planes(r7, qp, kvp, d["hw"], f16=True)
    feats, fxp = ops.ipa_attention_planes(q_xp, k_xp, v_vf, pts, bias, pz, mask, r7, f16=True)
    q, _ = lin(w["q"]); kv, _ = lin(w["kv"])
    q_pts, k_pts, v_pts = ops.ipa_prep_points(r7, qp.view(B, N, -1), kvp.view(B, N, -1), 8, 8, 12)
    ref = ops.ipa_attention(q.view(B, N, H, -1), kv.view(B, N, H, -1), q_pts, k_pts, v_pts, bias, pz, mask, r7, d["hw"]).view(M, -1)
    got = ops.unpack_planes(fxp, M, 2688)
    fr = fxp.view(torch.float16).reshape(-1, 168, 2, 2, 32, 8).float()
    for name, sl in (("o", slice(0, 2048)), ("o_pt", slice(2048, 2432)), ("o_pair", slice(2432, 2688))):
        gg = got if name == "o" else feats.view(M, -1)
        e = (gg[:, sl] - ref[:, sl]).abs()
        print(name, e.max().item() / ref[:, sl].abs().max().item())
    e = (got[:, :2048] - ref[:, :2048]).abs()
    idx = (e > 1e-5).nonzero()
    print("bad o elements:", len(idx), idx[:10].tolist(), "heads:", sorted(set((idx[:, 1] // 256).tolist())), "rows%64:", sorted(set((idx[:,0] % 64).tolist()))[:20])
    feats2, fxp2 = ops.ipa_attention_planes(q_xp, k_xp, v_vf, pts, bias, pz, mask, r7, f16=True)
    got2 = ops.unpack_planes(fxp2, M, 2688)
    print("run-to-run max diff in o:", (got2[:, :2048] - got[:, :2048]).abs().max().item())
    for r, c in idx[:6].tolist():
        print(r, c, "got", got[r, c].item(), "got2", got2[r, c].item(), "ref", ref[r, c].item())
    # decode v_vf (f16 pair) and compare with the fp32 v projection
    RT = M // 32; Nv = 2048; tph = 8
    frv = v_vf.view(torch.float16).reshape(RT, Nv // 32 // tph, tph, 2, 2, 2, 32, 8).float()
    for label, fr_ in (("h+l", frv.sum(4)), ("h only", frv[:, :, :, :, 0])):
        u = torch.arange(2)[:, None, None]; hh = torch.arange(2)[None, :, None]; j = torch.arange(8)[None, None, :]
        r = 8 * u + j
        row = ((r & 3) + 8 * (r >> 2) + 4 * hh).to(DEV)
        y = torch.zeros(RT, 32, Nv, device=DEV)
        cols = fr_.permute(0, 3, 4, 6, 1, 2, 5).reshape(RT, 2, 2, 8, Nv)
        y[:, row.reshape(-1)] = cols.reshape(RT, 32, Nv)
        y = y.reshape(-1, Nv)
        vref = kv.view(M, H, 2, 256)[:, :, 1].reshape(M, Nv)
        e = (y - vref).abs()
        print("v_vf decode", label, "max err", e.max().item(), "n > 1e-5:", (e > 1e-5).sum().item(), "of", e.numel())
        if label == "h+l":
            bad = (e > 1e-5).nonzero()[:6]
            for rr, cc in bad.tolist():
                print("  ", rr, cc, "dec", y[rr, cc].item(), "ref", vref[rr, cc].item())
    _, qb = lin(w["q"], want_f32=False, want_xp=True, xp_bf16=True)
    _, kb = lin(w["k"], want_f32=False, want_xp=True, xp_bf16=True)
    vb = ops.node_linear_vfrag(s_xp, w["v"]["w"], w["v"]["b"], M, 256, 2048, 8, f16=False)
    ptb = ops.ipa_prep_points_planes(r7, qp, kvp, d["hw"], f16=False)
    _, fxb = ops.ipa_attention_planes(qb, kb, vb, ptb, bias, pz, mask, r7, f16=False)
    gb = ops.unpack_planes(fxb, M, 2688)
    print("bf16-planes kernel o vs ref:", ((gb[:, :2048] - ref[:, :2048]).abs().max() / ref[:, :2048].abs().max()).item())
    for r, c in idx[:6].tolist():
        print(r, c, "f16 kernel", got[r, c].item(), "bf16 kernel", gb[r, c].item(), "ref", ref[r, c].item())
    # stored planes of the bad elements
    def planes_at(xp_, row, col, KS):
        rt, m = row // 32, row % 32
        t_, rem = col // 32, col % 32          # chain order: col = 32 (ks>>1) + (r&3) + 8 (r>>2) + 4 g, r = 8 (ks&1) + j
        g_ = (rem >> 2) & 1
        rr = (rem & 3) + 4 * (rem >> 3)
        ks = 2 * t_ + (rr >> 3); j_ = rr & 7
        fr_ = xp_.view(torch.float16).reshape(-1, KS, 2, 2, 32, 8)
        return fr_[rt, ks, 0, g_, m, j_].item(), fr_[rt, ks, 1, g_, m, j_].item()
    for r, c in idx[:6].tolist():
        print(r, c, "stored (h, l):", planes_at(fxp, r, c, 168), "ref", ref[r, c].item())
