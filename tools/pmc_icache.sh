#!/bin/bash
# tools/pmc_icache.sh <kernel-name-substring> -- <command...> : instruction-fetch counters of one kernel (is straight-line code fetch-bound?)
KN=$1; shift 2
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/pmc_icache; rm -rf $OUT; mkdir -p $OUT
i=0
for grp in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_WAVE_CYCLES" "SQ_IFETCH SQ_WAIT_INST_ANY SQ_BUSY_CYCLES SQ_WAVES" "SQC_ICACHE_MISSES_DUPLICATE SQ_INSTS_VALU SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_ANY"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/g$i -- "$@" > $OUT/g$i.log 2>&1 || tail -3 $OUT/g$i.log
done
python - <<PY
import csv, glob, collections
for g in sorted(glob.glob("$OUT/g*/")):
    for f in glob.glob(g + "**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if "$KN" in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, v in acc.items():
            print(f"{k:32s} n={len(v)} mean={sum(v)/len(v):.4g}")
PY
