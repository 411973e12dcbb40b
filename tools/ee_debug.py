import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
from conftest import T, golden, synth_sd
from str2str_amd.factory import build_synthetic_net
net = build_synthetic_net(device="cuda")
g = golden("embedding.npz")
DEV = "cuda"
a = net.embedder(residue_idx=T(g["residue_idx"]), t=T(g["t"]), fixed_mask=T(g["fixed_mask"]).to(DEV), self_conditioning_ca=T(g["sc_ca"]).to(DEV))
ipa0 = net.translator.trunk["ipa_0"]
b = net.embedder(residue_idx=T(g["residue_idx"]), t=T(g["t"]), fixed_mask=T(g["fixed_mask"]).to(DEV), self_conditioning_ca=T(g["sc_ca"]).to(DEV), next_proj=ipa0.pair_proj_weights())
e1, e2 = a[1], b[1]
d = (e1 - e2).abs()
print("shape", tuple(e1.shape), "max diff", d.max().item(), "n diff", (d > 0).sum().item(), "of", d.numel())
idx = (d > 0).nonzero()
print(idx[:20].tolist())
print("channels with diffs:", sorted(set(idx[:, 3].tolist()))[:40])
print("pairs with diffs:", len(set((idx[:, 0] * 10000 + idx[:, 1] * 100 + idx[:, 2]).tolist())))
ge = torch.tensor(g["edge"]).to(DEV)
print("plain vs golden max", (e1 - ge).abs().max().item(), " fused vs golden max", (e2 - ge).abs().max().item())
bad = ((e2 - ge).abs() > 1e-3).nonzero()
print("fused bad:", bad[:12].tolist(), len(bad))
bad = ((e1 - ge).abs() > 1e-3).nonzero()
print("plain bad:", bad[:12].tolist(), len(bad))
js = sorted(set((idx[:, 1] * 16 + idx[:, 2]).tolist()))
print("pair ids (i*16+j) with diffs, batch-agnostic:", js[:80])
