"""Timeline of one EdgeTransition workgroup (s_memtime stamps).  Builds a probe variant of the library next to the
real one (-DS2S_ET_PROBE=<block>) and prints per-stage durations and barrier waits for the 4 waves.

    python tools/et_probe.py build [block]      # on the CPU container (hipcc)
    python tools/et_probe.py run  [--B 16 --N 256]   # on the GPU box
"""
import ctypes
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "str2str_amd", "csrc", "build", "lib_etprobe.so")

if sys.argv[1] == "build":
    block = sys.argv[2] if len(sys.argv) > 2 else "3000"
    env = dict(os.environ, UNIT="pair_mlp_bf16")
    subprocess.run([os.path.join(ROOT, "tools", "build_variant.sh"), "etprobe", "-DS2S_ET_PROBE=" + block] + sys.argv[3:], check=True,
                   env=env, cwd=ROOT)
    sys.exit(0)

os.environ["STR2STR_HIP_LIB"] = LIB
os.environ["S2S_EDGE_MFMA"] = "bf16x6"
sys.path.insert(0, ROOT)
import argparse  # noqa: E402

import numpy as np  # noqa: E402
import torch  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("cmd")
ap.add_argument("--B", type=int, default=16)
ap.add_argument("--N", type=int, default=256)
a = ap.parse_args()
from str2str_amd import ops  # noqa: E402
from str2str_amd.factory import build_synthetic_net  # noqa: E402

net = build_synthetic_net(device="cuda")
et = net.translator.trunk["edge_transition_0"]
g = torch.Generator(device="cuda").manual_seed(0)
node = torch.randn(a.B, a.N, 256, device="cuda", generator=g)
edge = torch.randn(a.B, a.N, a.N, 128, device="cuda", generator=g)
mask = torch.ones(a.B, a.N, device="cuda")
with torch.no_grad():
    for _ in range(3):
        et(node, edge, edge_mask_1d=mask)
torch.cuda.synchronize()
lib = ops.load_library()
buf = np.zeros((4, 512), dtype=np.uint64)
rc = lib.s2s_debug_read_et_probe(ctypes.c_void_p(buf.ctypes.data))
assert rc == 0, rc
t = buf.astype(np.int64)
t0 = t[:, 0].min()
print("all times in shader cycles relative to the first wave's start")
print("wave  start  prologue_done  after_barrier0 |  loop_end  kernel_end")
for w in range(4):
    print(w, t[w, 0] - t0, t[w, 1] - t0, t[w, 2] - t0, "|", t[w, 100] - t0, t[w, 101] - t0)
print("stage: arrival at barrier (rel. to previous barrier release) / wait in barrier, per wave")
prev = t[:, 2].copy()
for s in range(29):
    arr, rel = t[:, 4 + 2 * s], t[:, 5 + 2 * s]
    print(f"{s:2d}", " ".join(f"{int(arr[w] - prev[w]):6d}/{int(rel[w] - arr[w]):5d}" for w in range(4)))
    prev = rel.copy()
print("tail (last barrier -> loop end):", [int(t[w, 100] - prev[w]) for w in range(4)])
print("epilogue:", [int(t[w, 101] - t[w, 100]) for w in range(4)])

if (t[:, 200] > 0).all():
    print("slot durations (cycles; 12 MFMAs = 384 ideal), waves 0..3")
    for sl in range(240):
        nxt = [t[w, 201 + sl] if sl < 239 else t[w, 100] for w in range(4)]
        print(f"slot {sl:3d} stage {sl // 8:2d}.{sl % 8}:", " ".join(f"{int(nxt[w] - t[w, 200 + sl]):6d}" for w in range(4)))
