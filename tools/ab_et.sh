#!/bin/bash
# tools/ab_et.sh <variant> <variant> ...: edge-transition (with the fused projection, cfg2 shape) and edge-embedding launch times of
# library variants (tools/build_variant.sh), interleaved A B C A B C in ONE call (box-to-box spread is larger than most effects).
for rep in 1 2; do
  for v in "$@"; do
    L=str2str_amd/csrc/build/ab_$v.so
    echo -n "$v: "; STR2STR_HIP_LIB=$L python tools/et_only.py --B 128 --N 256 --proj --iters 8 2>/dev/null | sed 's/.*: //'
  done
done
