set -x
mkdir -p gpurun_out/r02d
python -m pytest tests -m gpu -x -q 2>&1 | tail -8 > gpurun_out/r02d/pytest.log
python bench.py --steps 2 --warmup 1 > gpurun_out/r02d/bench_cfg2.json 2> gpurun_out/r02d/bench_cfg2.err
python bench.py --config cfg3 --steps 1 --warmup 0 > gpurun_out/r02d/bench_cfg3.json 2> gpurun_out/r02d/bench_cfg3.err
S2S_HIP_GRAPH=1 python bench.py --config cfg3 --steps 1 --warmup 0 > gpurun_out/r02d/bench_cfg3_graph.json 2> gpurun_out/r02d/bench_cfg3_graph.err
python bench.py --config cfg4 --steps 1 --warmup 0 --denoise-steps 40 > gpurun_out/r02d/bench_cfg4_s40.json 2> gpurun_out/r02d/bench_cfg4.err
python bench.py --config cfg5 --steps 1 --warmup 0 --denoise-steps 20 > gpurun_out/r02d/bench_cfg5_s20.json 2> gpurun_out/r02d/bench_cfg5.err
tail -3 gpurun_out/r02d/*.err
cat gpurun_out/r02d/pytest.log
for f in gpurun_out/r02d/bench_*.json; do echo $f; python -c "
import json,sys
l=json.loads(open('$f').read().strip().splitlines()[-1]); print(l['value'], l['ms_per_step'], l.get('roofline',{}).get('mean_launch_ms'), l.get('ipa_kernel',{}).get('mean_launch_ms'), l['config'].get('pdb_write_s'))"; done
