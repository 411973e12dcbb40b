"""Slot timeline of one edge-embedding workgroup (fourth tile of block <block>; -DS2S_ET_PROBE build of pair_mlp_bf16.hip).

    python tools/ee_probe.py build [block]        # CPU container (hipcc)
    python tools/ee_probe.py run [--B 16 --N 256]  # GPU box
"""
import ctypes
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "str2str_amd", "csrc", "build", "lib_eeprobe.so")

if sys.argv[1] == "build":
    block = sys.argv[2] if len(sys.argv) > 2 else "100"
    subprocess.run([os.path.join(ROOT, "tools", "build_variant.sh"), "eeprobe", "-DS2S_ET_PROBE=" + block] + sys.argv[3:], check=True,
                   env=dict(os.environ, UNIT="pair_mlp_bf16"), cwd=ROOT)
    sys.exit(0)

os.environ["STR2STR_HIP_LIB"] = LIB
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
B, N = a.B, a.N
g = torch.Generator(device="cuda").manual_seed(0)
idx = torch.arange(N)[None].repeat(B, 1)
t = torch.full((B,), 0.5)
ca = torch.randn(B, N, 3, device="cuda", generator=g) * 10
fixed = torch.zeros(B, N, device="cuda")
mask = torch.ones(B, N, device="cuda")
proj = net.translator.trunk["ipa_0"].pair_proj_weights()
with torch.no_grad():
    for _ in range(3):
        net.embedder(idx, t, fixed, ca, node_mask=mask, next_proj=proj)
torch.cuda.synchronize()
lib = ops.load_library()
buf = np.zeros((4, 512), dtype=np.uint64)
assert lib.s2s_debug_read_et_probe(ctypes.c_void_p(buf.ctypes.data)) == 0
tt = buf.astype(np.int64)
print("slot durations of one tile (cycles; 12 MFMAs = 384 ideal), waves 0..3")
tot = np.zeros(4, dtype=np.int64)
for s in range(40):
    d = tt[:, 301 + s] - tt[:, 300 + s]
    tot += d
    print(f"slot {s:2d} (stage {s // 8}.{s % 8}):", " ".join(f"{int(x):6d}" for x in d))
print("slots total:", tot, " projection store:", tt[:, 341] - tt[:, 340], " tile:", tt[:, 341] - tt[:, 300])
