# round-3 evidence run (one gpurun call): box calibration, GPU tests, the four BASELINE workloads, rocprof kernel stats of the cfg2
# step, launch sequence of one evaluation, PMC HBM traffic (pair kernels at B = 16, IPA pair at B = 128), PMC issue / stall counters of
# the two kernels the rooflines are quoted on.   bash tools/run_r03.sh <tag>
T=${1:-r03}
O=gpurun_out/$T; mkdir -p $O
python tools/et_only.py --B 128 --N 256 --iters 20 --proj --layout tiled 2>/dev/null | tail -1 > $O/box_calibration.txt
python -m pytest tests -m gpu -q 2>&1 | tail -4 > $O/pytest.log
python bench.py > $O/bench_cfg2.json 2> $O/bench_cfg2.err
python bench.py --config cfg3 --steps 1 --warmup 1 > $O/bench_cfg3.json 2> $O/bench_cfg3.err
python bench.py --config cfg4 --steps 1 --warmup 0 > $O/bench_cfg4.json 2> $O/bench_cfg4.err
python bench.py --config cfg5 --steps 1 --warmup 0 > $O/bench_cfg5.json 2> $O/bench_cfg5.err
bash tools/prof_bench.sh $T > $O/prof.log 2>&1
cat $O/box_calibration.txt $O/pytest.log
for c in cfg2 cfg3 cfg4 cfg5; do python - <<PY
import json
l=json.loads(open("$O/bench_$c.json").read().strip().splitlines()[-1])
print("$c", round(l["value"],3), "conf/s", round(l["ms_per_step"],1), "ms/step", "roofline", round(l.get("roofline",{}).get("frac") or 0,4), "ipa", round(l.get("ipa_kernel",{}).get("frac") or 0,4), l["config"].get("pdb_write_s"))
PY
done
head -14 gpurun_out/${T}_bench_kernel_stats.md
DB=$(ls gpurun_out/prof_$T/*/*results.db gpurun_out/prof_$T/*results.db 2>/dev/null | head -1)
python tools/rocpd_sequence.py $DB gpurun_out/${T}_eval_sequence.md > /dev/null
rm -rf gpurun_out/prof_$T
bash tools/pmc_hbm_traffic.sh gpurun_out/${T}_pmc_hbm_traffic.json 16 256 > $O/pmc.log 2>&1
rm -rf gpurun_out/pmc_traffic
bash tools/pmc_ipa.sh gpurun_out/${T}_pmc_ipa_traffic.json > $O/pmc_ipa.log 2>&1; tail -1 $O/pmc_ipa.log
bash tools/pmc_kernel.sh ${T}_et edge_transition_f16 -- python tools/et_only.py --B 64 --N 256 --iters 2 --proj --layout tiled > gpurun_out/${T}_pmc_et_f16_counters.txt 2>&1
bash tools/pmc_kernel.sh ${T}_ipa ipa_attention_f16w -- python tools/ipa_loop.py --seconds 0.5 > gpurun_out/${T}_pmc_ipa_f16w_counters.txt 2>&1
rm -rf gpurun_out/pmc_${T}_et gpurun_out/pmc_${T}_ipa
tail -22 gpurun_out/${T}_pmc_et_f16_counters.txt
