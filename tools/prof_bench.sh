#!/bin/bash
# rocprofv3 kernel trace of one bench step:  tools/prof_bench.sh <tag> [bench args]   -> gpurun_out/<tag>_bench_kernel_stats.{csv,md}
TAG=$1; shift
OUT=$PWD/gpurun_out
mkdir -p $OUT/prof_$TAG
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT/prof_$TAG -o bench -- python $OLDPWD/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-kernel-table "$@" > $OUT/${TAG}_bench_line_under_rocprof.json 2> $OUT/prof_$TAG.err
cd $OLDPWD
DB=$(ls $OUT/prof_$TAG/*/*results.db $OUT/prof_$TAG/*results.db 2>/dev/null | head -1)
python tools/rocpd_summary.py $DB $OUT/${TAG}_bench_kernel_stats | head -32
