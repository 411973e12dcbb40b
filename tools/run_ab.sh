mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | grep -v "^  File\|^Extension\|^$" | tail -4
python bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/bench_f16.json 2> gpurun_out/bench_f16.err
tail -1 gpurun_out/bench_f16.json | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['mean_launch_ms'], d['roofline']['frac'], d['ipa_kernel']['mean_launch_ms'])"
bash tools/prof_bench.sh r02j > gpurun_out/r02j_prof.log 2>&1
DB=$(ls gpurun_out/prof_r02j/*/*results.db gpurun_out/prof_r02j/*results.db 2>/dev/null | head -1)
python tools/rocpd_sequence.py $DB gpurun_out/r02j_eval_sequence.md
rm -rf gpurun_out/prof_r02j
head -24 gpurun_out/r02j_bench_kernel_stats.md
