mkdir -p gpurun_out/ipa6
timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -4
timeout 600 python tools/ipa_block_bench.py > gpurun_out/ipa6/cfg2.txt 2>&1; grep -A9 "planes path" gpurun_out/ipa6/cfg2.txt | grep -E "attention|total"
timeout 600 python tools/ipa_block_bench.py --N 512 --B 32 > gpurun_out/ipa6/n512.txt 2>&1; grep -E "attention|total" gpurun_out/ipa6/n512.txt
timeout 600 python tools/ipa_block_bench.py --N 96 --B 64 > gpurun_out/ipa6/n96.txt 2>&1; grep -E "attention|total|vs fp32" gpurun_out/ipa6/n96.txt
python bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/ipa6/bench_planes.json 2> gpurun_out/ipa6/bench_planes.err; tail -c 1500 gpurun_out/ipa6/bench_planes.json
S2S_IPA_PATH=f32 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/ipa6/bench_f32.json 2> gpurun_out/ipa6/bench_f32.err; tail -c 600 gpurun_out/ipa6/bench_f32.json
