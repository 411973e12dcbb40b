"""Phase timeline of one tile of the width-split edge transition (csrc/edge_transition_ws.hip, -DS2S_WS_PROBE):
    UNIT=edge_transition_ws bash tools/build_variant.sh wsprobe -DS2S_WS_PROBE
    S2S_ET_KERNEL=ws STR2STR_HIP_LIB=str2str_amd/csrc/build/ab_wsprobe.so python tools/ws_phase_probe.py
s_memtime stamps of thread 0 of workgroup 0 (100 MHz-independent shader clock) at the phase boundaries; prints the deltas of a few tiles."""
import ctypes
import os
import subprocess
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["S2S_ET_KERNEL"] = "ws"
from str2str_amd import ops  # noqa: E402

lib = ops.load_library()
import torch  # noqa: E402
from str2str_amd.factory import build_synthetic_net  # noqa: E402

net = build_synthetic_net(device="cuda")
et = net.translator.trunk["edge_transition_0"]
B, N = 128, 256
g = torch.Generator(device="cuda").manual_seed(0)
node = torch.randn(B, N, 256, device="cuda", generator=g)
edge = torch.randn(B, N, N, 128, device="cuda", generator=g)
mask = torch.ones(B, N, device="cuda")
n_p, node_ab = et.node_parts(ops.to_act(node.reshape(B * N, -1).contiguous(), "f16x3"), B * N)
zt = ops.pair_tiled(edge)
del edge
nxt = net.translator.trunk["ipa_1"].pair_proj_weights()
with torch.no_grad():
    for _ in range(2):
        et.pair_mlp(zt, node_ab.view(B, N, -1), n_p.view(B, N, -1), mask, nxt, out_layout="tiled")
    torch.cuda.synchronize()
buf = np.zeros(8 * 64, dtype=np.uint64)
lib.s2s_ws_probe_read.argtypes = [ctypes.c_void_p]
lib.s2s_ws_probe_read(buf.ctypes.data)
st = buf.reshape(8, 64).astype(np.int64)
names = {0: "tile start"}
for r in range(3):
    names[1 + 5 * r] = f"A{r} MFMAs done"
    names[2 + 5 * r] = f"A{r} epilogue done (planes in regs)"
    names[3 + 5 * r] = f"barrier (ring free)"
    names[4 + 5 * r] = f"ring written + barrier"
    names[5 + 5 * r if r < 2 else 15] = f"B{r} MFMAs done"
    names[16 + 4 * r] = f"F{r} epilogue done"
    names[17 + 4 * r] = "barrier (ring free)"
    names[18 + 4 * r] = "ring written + barrier"
    names[19 + 4 * r] = f"F{r} MFMAs done"
names.update({28: "LN local stats", 29: "LN barrier", 30: "LN normalise + store", 31: "x_store + ring_put + prefetch", 32: "proj barrier", 34: "proj MFMAs", 35: "next context + prefetch issue", 36: "pair-vector stores", 33: "projection stores"})
order_tail = [28, 29, 30, 31, 32, 34, 35, 36, 33]
order = [k for k in sorted(names) if k < 28] + order_tail
for it in (2, 3, 4):
    row = st[it]
    print(f"tile {it}: total {row[33] - row[0]} ticks")
    prev = row[0]
    for k in order[1:]:
        print(f"   {names[k]:42s} {row[k] - prev:8d}")
        prev = row[k]
