# round-2 evidence run: box calibration, GPU tests, the four BASELINE workloads, rocprof kernel stats of the cfg2 step
mkdir -p gpurun_out/r02h
python tools/et_only.py --B 128 --N 256 --iters 20 --proj 2>/dev/null | tail -1 > gpurun_out/r02h/box_calibration.txt
python -m pytest tests -m gpu -x -q 2>&1 | tail -4 > gpurun_out/r02h/pytest.log
python bench.py --steps 3 --warmup 1 > gpurun_out/r02h/bench_cfg2.json 2> gpurun_out/r02h/bench_cfg2.err
python bench.py --config cfg3 --steps 1 --warmup 0 > gpurun_out/r02h/bench_cfg3.json 2> gpurun_out/r02h/bench_cfg3.err
python bench.py --config cfg4 --steps 1 --warmup 0 > gpurun_out/r02h/bench_cfg4.json 2> gpurun_out/r02h/bench_cfg4.err
python bench.py --config cfg5 --steps 1 --warmup 0 > gpurun_out/r02h/bench_cfg5.json 2> gpurun_out/r02h/bench_cfg5.err
bash tools/prof_bench.sh r02h > gpurun_out/r02h/prof.log 2>&1
cat gpurun_out/r02h/box_calibration.txt gpurun_out/r02h/pytest.log
for c in cfg2 cfg3 cfg4 cfg5; do python - <<PY
import json
l=json.loads(open("gpurun_out/r02h/bench_$c.json").read().strip().splitlines()[-1])
print("$c", round(l["value"],3), "conf/s", round(l["ms_per_step"],1), "ms/step", l.get("roofline",{}).get("mean_launch_ms"), l.get("ipa_kernel",{}).get("mean_launch_ms"), l["config"].get("pdb_write_s"))
PY
done
head -14 gpurun_out/r02h_bench_kernel_stats.md
