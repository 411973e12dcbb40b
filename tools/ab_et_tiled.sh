#!/bin/bash
# tools/ab_et_tiled.sh <variant> ...: as tools/ab_et.sh, in the trunk's configuration (tiled pair tensor in and out, fused projection), cfg2's launch
for rep in 1 2 3; do
  for v in "$@"; do
    L=str2str_amd/csrc/build/ab_$v.so
    echo -n "$v: "; STR2STR_HIP_LIB=$L python tools/et_only.py --B 128 --N 256 --proj --layout tiled --iters 8 2>/dev/null | sed 's/.*: //'
  done
done
