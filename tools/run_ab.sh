mkdir -p gpurun_out
python tools/ipa_dbg.py 2>&1 | grep -E "^o |^o_|bad o|bf16-planes kernel o"
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | grep -v "^  File\|^Extension\|^$" | tail -4
python bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/bench_f16.json 2> gpurun_out/bench_f16.err
tail -1 gpurun_out/bench_f16.json | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['mean_launch_ms'], d['roofline']['frac'], d['ipa_kernel']['mean_launch_ms'], d['ipa_kernel']['frac'])"
