#!/bin/bash
# tools/pmc_kernel.sh <tag> <kernel-name-substring> -- <command...>   : stall / issue PMC groups for one kernel
TAG=$1; KN=$2; shift 3
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/pmc_$TAG; mkdir -p $OUT
i=0
for grp in "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES" \
           "GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU SQ_INSTS_MFMA SQ_BUSY_CYCLES SQ_ACTIVE_INST_LDS" \
           "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_SMEM" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/g$i -- "$@" > $OUT/g$i.log 2>&1
done
python - <<PY
import csv, glob, collections
dur = []
for g in sorted(glob.glob("$OUT/g*/")):
    for f in glob.glob(g + "**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if "$KN" in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, v in acc.items():
            print(f"{k:32s} n={len(v)} mean={sum(v)/len(v):.4g}")
    for f in glob.glob(g + "**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "$KN" in r["Kernel_Name"]:
                dur.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
print("kernel ms (profiled):", sum(dur) / max(1, len(dur)))
PY
