"""Build libstr2str_hip.so in-tree with hipcc for gfx950 (no torch headers involved).

    python -m str2str_amd.build [--force]

Each .hip translation unit is compiled to an object (cached by mtime) and linked into
``str2str_amd/libstr2str_hip.so``.  The per-residue geometry kernels are built with
-ffp-contract=off so that every arithmetic op rounds like one eager PyTorch op of the reference.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "build")
LIB = os.path.join(HERE, "libstr2str_hip.so")

ARCH = "gfx950"
COMMON = ["-O3", "-std=c++17", "-fPIC", f"--offload-arch={ARCH}", "-I", os.path.join(ROOT, "include"), "-I", CSRC,
          "-Wno-unused-result"]
UNITS = {
    "abi.hip": [],
    "pdb_format.cpp": [],   # host-only C++ (PDB text writer / merger behind the C ABI)
    "host_rng.cpp": [],     # host-only C++ (fast-forward of the CPU generator over the reference's discarded step draws)
    "rigid_kernels.hip": ["-ffp-contract=off"],
    "se3_step.hip": ["-ffp-contract=off"],
    "forward_marginal.hip": ["-ffp-contract=off"],
    "ensemble_metrics.hip": ["-ffp-contract=off"],
    # the MFMA chains are fully unrolled on purpose (accumulator tiles must be statically indexed)
    "pair_mlp.hip": ["-mllvm", "-pragma-unroll-threshold=10000000"],
    # (no SLP vectorisation in the split-f16 pair kernels: hipcc packs the LayerNorm / epilogue arithmetic into v_pk_*_f32, and a packed
    #  fp32 instruction issued between MFMAs costs 6-10 matrix-pipe cycles against 2 x 2.3 for the two plain ones it replaces:
    #  tools/ubench/mfma_valu_overlap.hip; -0.3 .. -1.1 % per edge-transition launch, same-call A/B)
    "pair_mlp_f16.hip": ["-mllvm", "-pragma-unroll-threshold=10000000", "-fno-slp-vectorize"],
    # (the same source as two more translation units, compiled side by side: short-chain edge transition; edge embedding)
    "pair_mlp_f16_b.hip": ["-mllvm", "-pragma-unroll-threshold=10000000", "-fno-slp-vectorize"],
    "pair_mlp_f16_c.hip": ["-mllvm", "-pragma-unroll-threshold=10000000", "-fno-slp-vectorize"],
    "ipa_attention.hip": ["-mllvm", "-pragma-unroll-threshold=10000000"],
    "ipa_attention_f16w.hip": ["-mllvm", "-pragma-unroll-threshold=10000000", "-fno-slp-vectorize"],   # (as above; -0.5 .. -1.5 % per IPA block)
    "node_gemm.hip": ["-mllvm", "-pragma-unroll-threshold=10000000", "-fno-slp-vectorize"],   # (as above: node layers -2.4 %)
    # contraction off: the packed-plane output must be the exact split of the SAME rounded value the fp32 output stores
    "enc_attention.hip": ["-mllvm", "-pragma-unroll-threshold=10000000", "-ffp-contract=off", "-fno-slp-vectorize"],   # (-5 %)
}


def hipcc() -> str:
    for c in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found (set HIPCC)")


def _stale(out: str, deps) -> bool:
    if not os.path.exists(out):
        return True
    t = os.path.getmtime(out)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = True) -> str:
    os.makedirs(OBJ, exist_ok=True)
    cc = hipcc()
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")] + [
        os.path.join(ROOT, "include", "str2str_hip.h"), os.path.abspath(__file__)]

    def compile_one(item):
        src, extra = item
        s = os.path.join(CSRC, src)
        o = os.path.join(OBJ, os.path.splitext(src)[0] + ".o")
        inc = [os.path.join(CSRC, "pair_mlp_f16.hip")] if src.startswith("pair_mlp_f16_") else []   # (units that include another unit's source)
        if force or _stale(o, [s] + inc + headers):
            if src.endswith(".cpp"):
                cmd = [cc, "-x", "c++", "-c", s, "-o", o] + [c for c in COMMON if not c.startswith("--offload-arch")] + extra
            else:
                cmd = [cc, "-x", "hip", "-c", s, "-o", o] + COMMON + extra
            if verbose:
                print(" ".join(cmd), flush=True)
            subprocess.run(cmd, check=True)
        return o

    with ThreadPoolExecutor(max_workers=min(8, len(UNITS))) as ex:
        objs = list(ex.map(compile_one, UNITS.items()))
    for f in os.listdir(OBJ):                    # objects of units that are no longer part of the library (earlier variants)
        if f.endswith(".o") and os.path.join(OBJ, f) not in objs:
            os.remove(os.path.join(OBJ, f))
    if force or _stale(LIB, objs):
        cmd = [cc, "-shared", "-fPIC", f"--offload-arch={ARCH}", "-o", LIB] + objs
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
