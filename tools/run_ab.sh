mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
S2S_EDGE_MFMA=bf16x6 python tools/ee_time.py 2>/dev/null
python tools/ee_time.py 2>/dev/null
python bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/bench_f16.json 2> gpurun_out/bench_f16.err
tail -1 gpurun_out/bench_f16.json | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['mean_launch_ms'], d['roofline']['frac'], d['ipa_kernel']['mean_launch_ms'])"
