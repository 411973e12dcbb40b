"""Timeline of one workgroup of the f16 attention kernel (cycle stamps per key tile).
    python tools/ipa_f16w_probe.py build [block]   (CPU container)      python tools/ipa_f16w_probe.py run   (GPU box)"""
import ctypes
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
D = os.path.join(ROOT, "str2str_amd", "csrc", "build")
LIB = os.path.join(D, "ab_ipa8probe.so")
if sys.argv[1] == "build":
    block = sys.argv[2] if len(sys.argv) > 2 else "2000"
    env = dict(os.environ, UNIT="ipa_attention_f16w")
    subprocess.run([os.path.join(ROOT, "tools", "build_variant.sh"), "ipa8probe", "-DS2S_IPA_PROBE=" + block] + sys.argv[3:], check=True,
                   env=env, cwd=ROOT)
    sys.exit(0)
os.environ["STR2STR_HIP_LIB"] = LIB
sys.argv = [sys.argv[0]] + ["--iters", "2"]
sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np  # noqa: E402
import ipa_block_bench  # noqa: E402,F401  (runs the kernel)
from str2str_amd import ops  # noqa: E402

buf = np.zeros((4, 128), dtype=np.uint64)
assert ops.load_library().s2s_debug_read_ipa8_probe(ctypes.c_void_p(buf.ctypes.data)) == 0
t = buf.astype(np.int64)
print("prologue (loads + first DMA):", [int(t[w, 127] - t[w, 126]) for w in range(4)])
print("phase 1 per key tile, cycles (waves 0..3):  QK | xs write + wait | barrier | dma + logits")
NT = 8
for k in range(NT + 1):
    o = 6 * k
    print(f"tile {k}: " + "  |  ".join(" ".join("%5d" % (t[w, o + j + 1] - t[w, o + j]) for j in range(4)) for w in range(4)))
print("phase 2 per key tile:  wait + barrier + dma | PV (+ next tile exp / split)")
for k in range(NT):
    o = 60 + 6 * k
    print(f"tile {k}: " + "  |  ".join("%5d %5d" % (t[w, o + 1] - t[w, o], t[w, o + 3] - t[w, o + 1]) for w in range(4)))
print("phase 1 total:", [int(t[w, 120] - t[w, 127]) for w in range(4)], " phase 2 total:", [int(t[w, 121] - t[w, 120]) for w in range(4)],
      " epilogue:", [int(t[w, 122] - t[w, 121]) for w in range(4)])
print("item start to item start:", [int(t[0, 101 + k] - t[0, 100 + k]) for k in range(15)])
print("(kernel start to first item:", int(t[0, 100] - t[0, 126]), ")")
print("phase-2 prologue (accumulator init, probabilities of tile 0):", [int(t[w, 123] - t[w, 120]) for w in range(4)],
      " last step end -> 121:", [int(t[w, 121] - t[w, 60 + 6 * (NT - 1) + 3]) for w in range(4)])
print("phase 2, end of step t -> start of step t + 1:", [[int(t[w, 60 + 6 * (k + 1)] - t[w, 60 + 6 * k + 3]) for k in range(NT - 1)] for w in range(2)])
print("phase 1, end of step t -> start of step t + 1:", [[int(t[w, 6 * (k + 1)] - t[w, 6 * k + 4]) for k in range(NT)] for w in range(2)])
