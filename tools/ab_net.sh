#!/bin/bash
# tools/ab_net.sh <variant> ...: per-kernel-family times of a short cfg2 step (10 denoise steps) for library variants built by
# tools/build_variant.sh, interleaved in ONE call.  ("cur" = the library in the tree)
for rep in 1 2; do
  for v in "$@"; do
    L=str2str_amd/csrc/build/ab_$v.so; [ $v == cur ] && L=str2str_amd/libstr2str_hip.so
    STR2STR_HIP_LIB=$L python bench.py --steps 1 --warmup 1 --denoise-steps 10 --no-cpu-baseline --no-other-configs 2>/dev/null | python -c "
import json,sys
l=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=l['kernel_times']
print('$v', 'ms/step', round(l['ms_per_step'],1), {n.replace('s2s_',''): round(v['total_ms'],1) for n,v in k.items() if isinstance(v, dict)})"
  done
done
