mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
python bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/bench_f16.json 2> gpurun_out/bench_f16.err
tail -1 gpurun_out/bench_f16.json | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['mean_launch_ms'], d['roofline']['frac'], d['ipa_kernel']['mean_launch_ms'])"
bash tools/pmc_hbm_traffic.sh gpurun_out/r02i_pmc_hbm_traffic.json 16 256 > gpurun_out/pmc.log 2>&1
python -c "
import json; d=json.load(open('gpurun_out/r02i_pmc_hbm_traffic.json'))
for k,v in d['kernels'].items(): print(k, round(v['bytes_per_pair_corrected'],1) if 'bytes_per_pair_corrected' in v else v)
"
