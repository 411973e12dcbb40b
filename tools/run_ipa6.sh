mkdir -p gpurun_out/ipa6
timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -2
python bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/ipa6/bench_planes.json 2> gpurun_out/ipa6/bench_planes.err
python - <<'PY'
import json
l=json.loads(open('gpurun_out/ipa6/bench_planes.json').read().strip().splitlines()[-1])
print('planes', l['value'], l['ms_per_step'], l['roofline']['mean_launch_ms'], l['ipa_kernel']['mean_launch_ms'], l['ipa_kernel']['frac'])
PY
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
