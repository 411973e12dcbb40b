#!/bin/bash
# instruction-fetch counters of one kernel:  tools/pmc_icache.sh <tag> <kernel-name-substring> -- <command...>
TAG=$1; KN=$2; shift 3
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/pmc_$TAG; mkdir -p $OUT
i=0
for grp in "SQ_IFETCH SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_INSTS_VALU" \
           "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE" \
           "SQ_IFETCH_LEVEL SQ_WAIT_IFETCH SQ_INST_LEVEL_LDS SQ_WAVES"; do
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
