#!/bin/bash
# PMC passes over the EdgeTransition kernel alone (one counter group per run; no trace domains besides kernel-trace).
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/pmc_et_$1; MODE=${2:-bf16x6}; mkdir -p $OUT
i=0
for grp in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE" \
           "SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAVE_CYCLES SQ_BUSY_CYCLES" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES" \
           "GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_SALU"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/g$i -- python tools/et_only.py --B 16 --N 256 --mode $MODE > $OUT/g$i.log 2>&1
done
python - <<PY
import csv, glob, collections
for g in sorted(glob.glob("$OUT/g*/")):
    for f in glob.glob(g + "**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if "edge_transition" in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, v in acc.items():
            print(f"{k:32s} n={len(v)} mean={sum(v)/len(v):.4g}")
PY
