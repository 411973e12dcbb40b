"""Where does a tile of the edge-transition kernel spend its cycles?  Needs the probe build:
    bash tools/build_variant.sh probe -DS2S_ET_PROBE=1
    STR2STR_HIP_LIB=str2str_amd/csrc/build/ab_probe.so python tools/et_phase_probe.py [--B 128]
s_memtime stamps at 16 points of a tile (wave 0 of every workgroup), differences summed per workgroup (csrc/pair_mlp_f16.hip,
S2S_ET_PROBE).  Prints cycles per tile and per slot of every phase (s_memtime counts at the 100 MHz-independent shader clock)."""
import argparse
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ap = argparse.ArgumentParser()
ap.add_argument("--B", type=int, default=128)
ap.add_argument("--N", type=int, default=256)
ap.add_argument("--fine", action="store_true", help="library built with -DS2S_ET_PROBE=2: slot tops 216 .. 239")
ap.add_argument("--block", action="store_true", help="library built with -DS2S_ET_PROBE=3: slot tops 72 .. 87 (layer-2 block B_4 A_6)")
ap.add_argument("--layout", default="tiled", choices=["rowmajor", "tiled", "none"], help="pair-tensor layout in (rowmajor|tiled) and out")
a = ap.parse_args()
from str2str_amd import ops  # noqa: E402
from str2str_amd.factory import build_synthetic_net  # noqa: E402

lib = ops.load_library()
lib.s2s_et_probe_read.argtypes = [ctypes.c_void_p, ctypes.c_int]
net = build_synthetic_net(device="cuda")
et = net.translator.trunk["edge_transition_0"]
kw = {"next_proj": net.translator.trunk["ipa_1"].pair_proj_weights()}
g = torch.Generator(device="cuda").manual_seed(0)
node = torch.randn(a.B, a.N, 256, device="cuda", generator=g)
edge = torch.randn(a.B, a.N, a.N, 128, device="cuda", generator=g)
mask = torch.ones(a.B, a.N, device="cuda")
n_p, node_ab = et.node_parts(ops.to_act(node.reshape(a.B * a.N, -1).contiguous(), "f16x3"), a.B * a.N, kernel_form=True)
n_p, node_ab = n_p.view(a.B, a.N, -1), node_ab.view(a.B, a.N, -1)
if a.layout != "rowmajor":
    edge = ops.pair_tiled(edge)
run = lambda: et.pair_mlp(edge, node_ab, n_p, mask, kw["next_proj"], out_layout=a.layout, ab_kernel_form=True)
buf = np.zeros(512 * 17, dtype=np.uint64)
with torch.no_grad():
    for _ in range(2):
        run()
    torch.cuda.synchronize()
    lib.s2s_et_probe_read(buf.ctypes.data, 1)
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(3):
        run()
    e.record()
    torch.cuda.synchronize()
    lib.s2s_et_probe_read(buf.ctypes.data, 1)
ms = s.elapsed_time(e) / 3
c = buf.reshape(512, 17).astype(np.float64)
c = c[c[:, 16] > 0]
tiles = c[:, 16].sum()
per = c[:, :15].sum(0) / tiles
names = [("A0 + previous tile's LayerNorm sums (4 slots)", 4), ("A1 + deviations, first normalised pieces (4 slots)", 4),
         ("P of the previous tile + its normalise / store / split (8 slots)", 8), ("B0 A2 (16 slots; first: projection stores)", 16),
         ("B1 A3 .. B9 A11 (144 slots)", 144), ("B10 (11 slots + MFMAs of slot 179)", 12), ("exposed: tile 11 -> planes", 0),
         ("B11 (11 slots + MFMAs of 191)", 12), ("-", 0), ("F k-steps 0-7 (15 slots + MFMAs of 207)", 16), ("-", 0),
         ("F k-steps 8-15 (15 slots + MFMAs of 223)", 16), ("-", 0), ("F k-steps 16-23 (15 slots + MFMAs of 239)", 16), ("end of the pass", 0)]
if a.fine:
    names = [(f"slots {x} .. {y - 1}", y - x) for x, y in zip([216, 220, 224, 225, 226, 227, 228, 229, 230, 232, 234, 236, 237, 238], [220, 224, 225, 226, 227, 228, 229, 230, 232, 234, 236, 237, 238, 239])]
    names.append(("slot 239 up to its exposed step", 1))
if a.block:
    names = [(f"slot {x} ({'B_4 k-step %d tiles %d,%d' % ((x - 72) // 6, 2 * ((x - 72) % 6), 2 * ((x - 72) % 6) + 1) if x < 84 else 'A_6 k-steps %d,%d' % (2 * (x - 84), 2 * (x - 84) + 1)})", 1) for x in range(72, 87)]
tot = per.sum()
print(f"{ms:.3f} ms/launch, {int(tiles)} probed tiles on {len(c)} workgroups; {tot:.0f} counter ticks per tile (stamp to stamp)")
for (n, slots), v in zip(names, per):
    print(f"  {n:52s} {v:9.0f} ticks  {100 * v / tot:5.1f} %" + (f"   {v / slots:6.1f} / slot" if slots else ""))
