mkdir -p gpurun_out/final
S=$(date +%s); python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2; echo "smoke s: $(( $(date +%s) - S ))"
S=$(date +%s); python bench.py > gpurun_out/final/bench_default.json 2> gpurun_out/final/bench_default.err; echo "default bench s: $(( $(date +%s) - S ))"
tail -1 gpurun_out/final/bench_default.json | python -c "import json,sys; d=json.loads(sys.stdin.read()); print({k: d[k] for k in ('metric','value','unit','n_gpus','steps','warmup','ms_per_step','dtype','vs_baseline')}); print(d['roofline']['frac'], d['ipa_kernel']['frac'], d['cpu_baseline']['value'])"
