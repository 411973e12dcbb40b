#!/bin/bash
# emit gfx950 assembly of one unit and print register / scratch / size metadata:  tools/asm_stats.sh pair_mlp_bf16
U=$1; OUT=${2:-/root/repo/gpurun_out/asm/$U.s}
mkdir -p $(dirname $OUT)
FLAGS=$(python -c "from str2str_amd.build import UNITS, COMMON; print(' '.join(c for c in COMMON + UNITS['$U.hip'] if c != '-fPIC'))")   # the unit's flags of str2str_amd/build.py
hipcc -x hip -S --cuda-device-only str2str_amd/csrc/$U.hip -o $OUT $FLAGS -w $EXTRA
grep -E "^\s+\.(sgpr_count|vgpr_count|agpr_count|private_segment_fixed_size|group_segment_fixed_size|name):" $OUT | paste - - - - - - | sed 's/  */ /g'
grep -E "; codeLenInByte|; ScratchSize|; Occupancy" $OUT
