# round-2 evidence run: box calibration, GPU tests, the four BASELINE workloads, rocprof kernel stats of the cfg2 step
mkdir -p gpurun_out/r02m
python tools/et_only.py --B 128 --N 256 --iters 20 --proj --mode f16x3 2>/dev/null | tail -1 > gpurun_out/r02m/box_calibration.txt
python -m pytest tests -m gpu -x -q 2>&1 | tail -4 > gpurun_out/r02m/pytest.log
python bench.py --steps 3 --warmup 1 > gpurun_out/r02m/bench_cfg2.json 2> gpurun_out/r02m/bench_cfg2.err
python bench.py --config cfg3 --steps 1 --warmup 0 > gpurun_out/r02m/bench_cfg3.json 2> gpurun_out/r02m/bench_cfg3.err
python bench.py --config cfg4 --steps 1 --warmup 0 > gpurun_out/r02m/bench_cfg4.json 2> gpurun_out/r02m/bench_cfg4.err
python bench.py --config cfg5 --steps 1 --warmup 0 > gpurun_out/r02m/bench_cfg5.json 2> gpurun_out/r02m/bench_cfg5.err
bash tools/prof_bench.sh r02m > gpurun_out/r02m/prof.log 2>&1
cat gpurun_out/r02m/box_calibration.txt gpurun_out/r02m/pytest.log
for c in cfg2 cfg3 cfg4 cfg5; do python - <<PY
import json
l=json.loads(open("gpurun_out/r02m/bench_$c.json").read().strip().splitlines()[-1])
print("$c", round(l["value"],3), "conf/s", round(l["ms_per_step"],1), "ms/step", l.get("roofline",{}).get("mean_launch_ms"), l.get("ipa_kernel",{}).get("mean_launch_ms"), l["config"].get("pdb_write_s"))
PY
done
head -14 gpurun_out/r02m_bench_kernel_stats.md
DB=$(ls gpurun_out/prof_r02m/*/*results.db gpurun_out/prof_r02m/*results.db 2>/dev/null | head -1)
python tools/rocpd_sequence.py $DB gpurun_out/r02m_eval_sequence.md
rm -rf gpurun_out/prof_r02m
bash tools/pmc_hbm_traffic.sh gpurun_out/r02m_pmc_hbm_traffic.json 16 256 > gpurun_out/r02m/pmc.log 2>&1
rm -rf gpurun_out/pmc_traffic
bash tools/pmc_ipa_planes.sh gpurun_out/r02m_pmc_ipa_traffic.json 128 256 > gpurun_out/r02m/pmc_ipa.log 2>&1
rm -rf gpurun_out/pmc_ipa_planes
python -c "
import json; d=json.load(open('gpurun_out/r02m_pmc_ipa_traffic.json'))
print({k: round(v['ratio_to_algorithmic'],3) for k,v in d.items() if k.startswith('attention_plus')})
"
