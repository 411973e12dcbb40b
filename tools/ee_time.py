"""HIP-event time of the edge-embedding launch (with the fused projection) at the cfg2 shape, library from STR2STR_HIP_LIB."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from str2str_amd.factory import build_synthetic_net
B, N = int(sys.argv[1]) if len(sys.argv) > 1 else 128, int(sys.argv[2]) if len(sys.argv) > 2 else 256
net = build_synthetic_net(device="cuda")
g = torch.Generator(device="cuda").manual_seed(0)
idx = torch.arange(N)[None].repeat(B, 1)
ca = torch.randn(B, N, 3, device="cuda", generator=g) * 10
fixed = torch.zeros(B, N, device="cuda")
mask = torch.ones(B, N, device="cuda")
t_emb = net.embedder.time_embed(torch.full((1,), 0.5)).to("cuda")
proj = net.translator.trunk["ipa_0"].pair_proj_weights()
layout = os.environ.get("EE_LAYOUT", "rowmajor")   # rowmajor | tiled (what the network passes to its first EdgeTransition)
run = lambda: net.embedder(idx, None, fixed, ca, node_mask=mask, next_proj=proj, t_emb=t_emb, edge_layout=layout)
with torch.no_grad():
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    iters = int(os.environ.get("EE_ITERS", "10"))
    for _ in range(iters):
        run()
    e1.record()
    torch.cuda.synchronize()
print(os.environ.get("STR2STR_HIP_LIB", "main").split("/")[-1], layout, "embedder ms:", round(e0.elapsed_time(e1) / iters, 3))
