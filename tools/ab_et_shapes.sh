#!/bin/bash
# tools/ab_et_shapes.sh <variant> ...: edge-transition launch times at cfg2's launch and at small-chain launches (cfg3 / the default block); tiled + projection
for shape in "128 256" "1000 35" "1000 80" "100 80"; do
  B=${shape% *}; N=${shape#* }
  for rep in 1 2; do
    for v in "$@"; do
      L=str2str_amd/csrc/build/ab_$v.so
      echo -n "B=$B N=$N $v: "; STR2STR_HIP_LIB=$L python tools/et_only.py --B $B --N $N --proj --layout tiled --iters 10 2>/dev/null | sed 's/.*: //; s/ fp32.*//'
    done
  done
done
