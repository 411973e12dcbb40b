# closing evidence run of round 4 (one gpurun call, ~25 min): box calibration, GPU tests, smoke, the default bench line, rocprof kernel
# stats + launch sequence of the cfg2 step, and the PMC counter groups that were still missing: the edge embedding
# and the attention kernels cfg3 launches (short-chain form at N = 35, ragged streaming form at N = 80; B = 1000; item 4).
#   bash tools/run_r04_final.sh <tag>
T=${1:-r04n}
O=gpurun_out/$T; mkdir -p $O
python tools/et_only.py --B 128 --N 256 --iters 20 --proj --layout tiled 2>/dev/null | tail -1 > $O/box_calibration.txt
python -m pytest tests -m gpu -q 2>&1 | tail -4 > $O/pytest.log
python __graft_entry__.py --smoke 2>&1 | tail -1 > $O/smoke.txt
python bench.py > $O/bench_cfg2.json 2> $O/bench_cfg2.err
bash tools/prof_bench.sh $T > $O/prof.log 2>&1
cat $O/box_calibration.txt $O/pytest.log $O/smoke.txt
python - <<PY
import json
l=json.loads(open("$O/bench_cfg2.json").read().strip().splitlines()[-1])
print("other_configs", {k: round(v.get("value", 0), 2) for k, v in l.get("other_configs", {}).items()}); print("cfg2", round(l["value"],3), "conf/s", round(l["ms_per_step"],1), "ms/step", "roofline", round(l.get("roofline",{}).get("frac") or 0,4), "ipa", round(l.get("ipa_kernel",{}).get("frac") or 0,4))
PY
head -14 gpurun_out/${T}_bench_kernel_stats.md
DB=$(ls gpurun_out/prof_$T/*/*results.db gpurun_out/prof_$T/*results.db 2>/dev/null | head -1)
python tools/rocpd_sequence.py $DB gpurun_out/${T}_eval_sequence.md > /dev/null
rm -rf gpurun_out/prof_$T
EE_LAYOUT=tiled EE_ITERS=2 bash tools/pmc_kernel.sh ${T}_ee edge_embed_f16 -- python tools/ee_time.py 64 256 > gpurun_out/${T}_pmc_ee_f16_counters.txt 2>&1
bash tools/pmc_kernel.sh ${T}_ipa35 ipa_attention -- python tools/ipa_fold_ab.py --B 1000 --N 35 --iters 2 --only-folded > gpurun_out/${T}_pmc_ipa_short_n35_counters.txt 2>&1
bash tools/pmc_kernel.sh ${T}_ipa80 ipa_attention -- python tools/ipa_fold_ab.py --B 1000 --N 80 --iters 2 --only-folded > gpurun_out/${T}_pmc_ipa_ragged_n80_counters.txt 2>&1
rm -rf gpurun_out/pmc_${T}_ee gpurun_out/pmc_${T}_ipa35 gpurun_out/pmc_${T}_ipa80
tail -22 gpurun_out/${T}_pmc_ee_f16_counters.txt
tail -22 gpurun_out/${T}_pmc_ipa_ragged_n80_counters.txt
