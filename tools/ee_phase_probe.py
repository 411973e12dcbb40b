"""Where does a tile of the edge-embedding kernel spend its cycles?  Needs the probe build:
    bash tools/build_variant.sh eeprobe -DS2S_EE_PROBE=1
    STR2STR_HIP_LIB=str2str_amd/csrc/build/ab_eeprobe.so python tools/ee_phase_probe.py
(s_memtime stamps at the tops of slots 0, 4, .. of a 40-slot tile, wave 0 of every workgroup; see tools/et_phase_probe.py)."""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from str2str_amd import ops  # noqa: E402
from str2str_amd.factory import build_synthetic_net  # noqa: E402

B, N = 128, 256
BASE = int(os.environ.get("EE_PROBE_BASE", "-1"))   # >= 0: library built with -DS2S_EE_PROBE=2 -DS2S_EE_PROBE_BASE=<base>: slots base .. base + 13
lib = ops.load_library()
lib.s2s_et_probe_read.argtypes = [ctypes.c_void_p, ctypes.c_int]
net = build_synthetic_net(device="cuda")
g = torch.Generator(device="cuda").manual_seed(0)
idx = torch.arange(N)[None].repeat(B, 1)
ca = torch.randn(B, N, 3, device="cuda", generator=g) * 10
fixed = torch.zeros(B, N, device="cuda")
mask = torch.ones(B, N, device="cuda")
t_emb = net.embedder.time_embed(torch.full((1,), 0.5)).to("cuda")
proj = net.translator.trunk["ipa_0"].pair_proj_weights()
run = lambda: net.embedder(idx, None, fixed, ca, node_mask=mask, next_proj=proj, t_emb=t_emb, edge_layout="tiled")
buf = np.zeros(512 * 17, dtype=np.uint64)
with torch.no_grad():
    for _ in range(2):
        run()
    torch.cuda.synchronize()
    lib.s2s_et_probe_read(buf.ctypes.data, 1)
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    lib.s2s_et_probe_read(buf.ctypes.data, 1)
c = buf.reshape(512, 17).astype(np.float64)
c = c[c[:, 16] > 0]
tiles = c[:, 16].sum()
per = c[:, :14].sum(0) / tiles
names = [("layer 2, slots 0-3", 4), ("layer 2, slots 4-7", 4), ("layer 2, slots 8-11", 4), ("layer 2, slots 12-15", 4), ("layer 3, slots 16-19", 4),
         ("layer 3, slots 20-23", 4), ("layer 3, slots 24-27", 4), ("layer 3, slots 28-30", 3), ("slot 31 up to its exposed step", 1),
         ("exposed: LayerNorm statistics + first piece", 0), ("projection, slots 32-35", 4), ("projection, slots 36-38", 3), ("slot 39", 1),
         ("projection stores", 0)]
if BASE >= 0:
    names = [(f"slot {BASE + k}", 1) for k in range(14)]
tot = per.sum()
print(f"{int(tiles)} probed tiles; {tot:.0f} counter ticks per tile (40 slots: {40 * 192} at the matrix pipe's rate)")
for (n, slots), v in zip(names, per):
    print(f"  {n:48s} {v:9.0f} ticks  {100 * v / tot:5.1f} %" + (f"   {v / slots:6.1f} / slot" if slots else ""))
