"""Per-kernel timing on the GPU box (HIP events on the launch stream).

    python tools/kernel_bench.py --B 16 --N 256 [--iters 5]
Prints ms per launch and the achieved fp32 FLOP/s or GB/s against the algorithmic counts of
SURVEY.md §8(d) / DESIGN.md.
"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def timeit(fn, iters, warmup=2):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--B", type=int, default=16)
    ap.add_argument("--N", type=int, default=256)
    ap.add_argument("--iters", type=int, default=5)
    a = ap.parse_args()
    from str2str_amd import ops
    from str2str_amd.factory import build_synthetic_net

    dev = "cuda"
    B, N = a.B, a.N
    net = build_synthetic_net(device=dev)
    tr = net.translator.trunk
    g = torch.Generator(device=dev).manual_seed(0)
    node = torch.randn(B, N, 256, device=dev, generator=g)
    edge = torch.randn(B, N, N, 128, device=dev, generator=g)
    mask = torch.ones(B, N, device=dev)
    q = torch.randn(B, N, 4, device=dev, generator=g)
    r7 = torch.cat([q / q.norm(dim=-1, keepdim=True), torch.randn(B, N, 3, device=dev, generator=g)], -1).contiguous()
    res = {}
    pairs = B * N * N

    # the EdgeTransition launches of one network evaluation (f16x3): two with the pair tensor tiled in and out, the last one without
    # pair output, all three with the next block's fused projections -- (2 x 1184 + 672) / 3 algorithmic bytes per pair and launch
    et = tr["edge_transition_0"]
    if et.arith == "f16x3":
        n_p, node_ab = et.node_parts(ops.to_act(node.reshape(B * N, -1).contiguous(), "f16x3"), B * N, kernel_form=True)
        n_p, node_ab, nxt = n_p.view(B, N, -1), node_ab.view(B, N, -1), tr["ipa_1"].pair_proj_weights()
        zt = ops.pair_tiled(edge)

        def trunk_mix():
            et.pair_mlp(zt, node_ab, n_p, mask, nxt, out_layout="tiled", ab_kernel_form=True)
            et.pair_mlp(zt, node_ab, n_p, mask, nxt, out_layout="tiled", ab_kernel_form=True)
            et.pair_mlp(zt, node_ab, n_p, mask, nxt, out_layout="none", ab_kernel_form=True)

        ms = timeit(trunk_mix, a.iters) / 3
        del zt
        res["edge_transition"] = dict(ms=ms, tflops=pairs * 491520 / ms / 1e9, gbs=pairs * (2 * 1184 + 672) / 3 / ms / 1e6)
    else:
        ms = timeit(lambda: et(node, edge, edge_mask_1d=mask), a.iters)
        res["edge_transition"] = dict(ms=ms, tflops=pairs * 491520 / ms / 1e9, gbs=pairs * 1024 / ms / 1e6)

    ipa = tr["ipa_0"]
    d = ipa._derived()
    ms = timeit(lambda: ops.pair_project(edge, d["wp"], d["b64"]), a.iters)
    res["pair_project"] = dict(ms=ms, gbs=pairs * (512 + 160) / ms / 1e6)
    bias, pz = ops.pair_project(edge, d["wp"], d["b64"])
    qq, kv = ipa.linear_q(node).contiguous(), ipa.linear_kv(node).contiguous()
    qp, kp, vp = ops.ipa_prep_points(r7, ipa.linear_q_points(node).contiguous(), ipa.linear_kv_points(node).contiguous())
    ms = timeit(lambda: ops.ipa_attention(qq, kv, qp, kp, vp, bias, pz, mask, r7, d["hw"]), a.iters)
    bytes_ipa = B * 4 * (9512 * N + 40 * N * N)
    res["ipa_attention"] = dict(ms=ms, tflops=B * 9856 * N * N / ms / 1e9, gbs=bytes_ipa / ms / 1e6)
    ms = timeit(lambda: ipa(node, edge, None, mask, _rigids7=r7), a.iters)
    res["ipa_module_total"] = dict(ms=ms)

    emb = net.embedder
    idx = torch.arange(N)[None].repeat(B, 1)
    t = torch.full((B,), 0.5)
    ca = torch.randn(B, N, 3, device=dev, generator=g) * 10
    fixed = torch.zeros(B, N, device=dev)
    ms = timeit(lambda: emb(idx, t, fixed, ca, node_mask=mask), a.iters)
    res["embedding_total"] = dict(ms=ms, gbs=pairs * 512 / ms / 1e6, tflops=pairs * 65536 / ms / 1e9)

    batch = dict(residue_mask=mask.double(), fixed_mask=fixed.double(), residue_idx=idx, t=t, sc_ca_t=ca, rigids_t=r7,
                 torsion_angles_sin_cos=torch.zeros(B, N, 7, 2, device=dev, dtype=torch.float64),
                 aatype=torch.zeros(B, N, dtype=torch.long, device=dev))
    net.backbone_in_forward = False
    with torch.no_grad():
        ms = timeit(lambda: net(batch), max(2, a.iters // 2))
    F = 2251264 * N * N + 3.24e7 * N
    res["net_forward"] = dict(ms=ms, tflops_alg=B * F / ms / 1e9, conf_per_s_at_100_steps=B / (101 * ms / 1e3))

    from str2str_amd.factory import build_diffuser
    diff = build_diffuser("/tmp/str2str_cache")
    p8 = diff.step_params(t).to(dev)
    ms = timeit(lambda: diff.step(r7, r7, p8, 0.01, mask, mask), a.iters)
    res["se3_step"] = dict(ms=ms)
    print(json.dumps(dict(B=B, N=N, **{k: {kk: round(vv, 4) for kk, vv in v.items()} for k, v in res.items()}), indent=1))


if __name__ == "__main__":
    main()
