mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
python bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/bench_ee.json 2> gpurun_out/bench_ee.err
cat gpurun_out/bench_ee.json | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline'])"
