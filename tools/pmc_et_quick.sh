#!/bin/bash
# two PMC passes (stall buckets + clock) over the EdgeTransition kernel of the library in $STR2STR_HIP_LIB
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/pmc_etq_$1; mkdir -p $OUT; B=${2:-64}
i=0
for grp in "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES" \
           "GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU SQ_INSTS_MFMA SQ_BUSY_CYCLES"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/g$i -- python tools/et_only.py --B $B --N 256 > $OUT/g$i.log 2>&1
done
python - <<PY
import csv, glob, collections
dur = []
for g in sorted(glob.glob("$OUT/g*/")):
    for f in glob.glob(g + "**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if "edge_transition" in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, v in acc.items():
            print(f"{k:32s} n={len(v)} mean={sum(v)/len(v):.4g}")
    for f in glob.glob(g + "**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "edge_transition" in r["Kernel_Name"]:
                dur.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
print("kernel ms (profiled):", sum(dur) / max(1, len(dur)))
PY
