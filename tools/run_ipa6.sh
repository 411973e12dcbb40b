mkdir -p gpurun_out/ipa6
timeout 900 python -m pytest tests -m gpu -x -q -k "ipa" 2>&1 | tail -3
timeout 600 python tools/ipa_block_bench.py > gpurun_out/ipa6/cfg2.txt 2>&1; grep -A9 "planes path" gpurun_out/ipa6/cfg2.txt; tail -3 gpurun_out/ipa6/cfg2.txt
timeout 600 python tools/ipa_planes_probe.py run > gpurun_out/ipa6/probe.txt 2>&1; tail -12 gpurun_out/ipa6/probe.txt
