mkdir -p gpurun_out/ipa6
timeout 1200 python -m pytest tests -m gpu -x -q -k "ipa or golden or free" 2>&1 | tail -4
timeout 600 python tools/ipa_block_bench.py > gpurun_out/ipa6/cfg2.txt 2>&1; grep -A9 "planes path" gpurun_out/ipa6/cfg2.txt | grep -E "attention|total"
timeout 600 python tools/ipa_block_bench.py --N 64 --B 64 > gpurun_out/ipa6/n64.txt 2>&1; grep -E "attention|total|vs fp32" gpurun_out/ipa6/n64.txt
timeout 600 python tools/ipa_block_bench.py --N 32 --B 64 > gpurun_out/ipa6/n32.txt 2>&1; grep -E "vs fp32" gpurun_out/ipa6/n32.txt
timeout 600 python tools/ipa_planes_probe.py run > gpurun_out/ipa6/probe.txt 2>&1; grep -E "item start|phase 1 total" gpurun_out/ipa6/probe.txt
