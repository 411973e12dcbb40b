"""Timeline of one IPA attention workgroup (s_memtime stamps per key tile).
    python tools/ipa_probe.py build [block]   (CPU container)      python tools/ipa_probe.py run   (GPU box)"""
import ctypes
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
D = os.path.join(ROOT, "str2str_amd", "csrc", "build")
LIB = os.path.join(D, "lib_ipaprobe.so")
if sys.argv[1] == "build":
    block = sys.argv[2] if len(sys.argv) > 2 else "700"
    env = dict(os.environ, UNIT="ipa_attention")
    subprocess.run([os.path.join(ROOT, "tools", "build_variant.sh"), "ipaprobe", "-DS2S_IPA_PROBE=" + block] + sys.argv[3:], check=True, env=env, cwd=ROOT)
    sys.exit(0)
os.environ["STR2STR_HIP_LIB"] = LIB
sys.argv = [sys.argv[0]] + ["--iters", "2"]
sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np  # noqa: E402
import ipa_only  # noqa: E402,F401  (runs the kernel)
from str2str_amd import ops  # noqa: E402

buf = np.zeros((4, 128), dtype=np.uint64)
assert ops.load_library().s2s_debug_read_ipa_probe(ctypes.c_void_p(buf.ctypes.data)) == 0
t = buf.astype(np.int64)
print("per key tile, cycles (waves 0..3):  QK+points | dma-issue | logits/softmax | PV | wait dma | barrier")
for k in range(8):
    o = 8 * k
    rows = []
    for w in range(4):
        rows.append("%5d %6d %6d %6d %5d %5d" % (t[w, o + 1] - t[w, o], t[w, o + 2] - t[w, o + 1], t[w, o + 3] - t[w, o + 2],
                                                  t[w, o + 4] - t[w, o + 3], t[w, o + 5] - t[w, o + 4], t[w, o + 6] - t[w, o + 5]))
    print(f"tile {k}: " + "  |  ".join(rows))
print("softmax section split: logits math | logits store | exp+sum | rescale (waves 0..3)")
for k in range(8):
    o, q = 8 * k, 64 + 4 * k
    print(f"tile {k}: " + "  |  ".join("%6d %6d %6d %6d" % (t[w, q] - t[w, o + 2], t[w, q + 1] - t[w, q], t[w, q + 2] - t[w, q + 1],
                                                             t[w, o + 3] - t[w, q + 2]) for w in range(4)))
print("loop total:", [int(t[w, 120] - t[w, 0]) for w in range(4)], " epilogue:", [int(t[w, 121] - t[w, 120]) for w in range(4)])
