"""Stress of the edge transition's weight ring (17 barriers per tile): many launches over random data and shapes, output checksums to a
file; run once with the tree's library and once with a -DS2S_ET_ALL_BARRIERS build (STR2STR_HIP_LIB) and compare the files -- a missing
barrier shows as a checksum that differs (or varies from repetition to repetition).
    python tools/et_ring_stress.py out.txt [--reps 3]"""
import argparse
import hashlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ap = argparse.ArgumentParser()
ap.add_argument("out")
ap.add_argument("--reps", type=int, default=3)
a = ap.parse_args()
from str2str_amd import ops  # noqa: E402
from str2str_amd.factory import build_synthetic_net  # noqa: E402

net = build_synthetic_net(device="cuda")
tr = net.translator.trunk
lines = []
with torch.no_grad():
    for case, (B, N) in enumerate([(128, 256), (16, 256), (1000, 35), (100, 80), (3, 47), (64, 128), (7, 300), (1, 512)]):
        g = torch.Generator(device="cuda").manual_seed(100 + case)
        node = torch.randn(B, N, 256, device="cuda", generator=g)
        edge = torch.randn(B, N, N, 128, device="cuda", generator=g)
        mask = (torch.rand(B, N, device="cuda", generator=g) > 0.05).float()
        for blk in (0, 1):
            et = tr[f"edge_transition_{blk}"]
            n_p, node_ab = et.node_parts(ops.to_act(node.reshape(B * N, -1).contiguous(), "f16x3"), B * N, kernel_form=True)
            zt = ops.pair_tiled(edge)
            proj = tr[f"ipa_{blk + 1}"].pair_proj_weights()
            sums = []
            for rep in range(a.reps):
                z, bias, pz = et.pair_mlp(zt, node_ab.view(B, N, -1), n_p.view(B, N, -1), mask, proj, out_layout="tiled", ab_kernel_form=True)
                h = hashlib.sha1()
                for t in (ops.pair_untiled(z), bias, pz):   # (the valid pairs: the padding of a partial last tile is never written)
                    h.update(t.cpu().numpy().tobytes())
                sums.append(h.hexdigest()[:16])
            lines.append(f"B={B} N={N} block={blk}: " + " ".join(sums))
            print(lines[-1], flush=True)
open(a.out, "w").write("\n".join(lines) + "\n")
