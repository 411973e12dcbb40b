mkdir -p gpurun_out/final
date +%s > gpurun_out/final/t0
true
python bench.py > gpurun_out/final/bench_default.json 2> gpurun_out/final/bench_default.err;  python - <<'PY'
import json
l=json.loads(open('gpurun_out/final/bench_default.json').read().strip().splitlines()[-1])
print({k:l[k] for k in ('metric','value','unit','n_gpus','steps','warmup','ms_per_step','higher_is_better','scaling','vs_baseline','dtype','data')})
print(l['roofline']['frac'], l['ipa_kernel']['frac'], l['cpu_baseline']['value'], l['cpu_baseline']['cores'])
PY
echo elapsed $(( $(date +%s) - $(cat gpurun_out/final/t0) )) s
