export GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$PWD}
bash tools/pmc_kernel.sh r02_ipa_planes ipa_attention_planes_kernel -- python tools/ipa_block_bench.py --iters 3 > gpurun_out/r02_pmc_ipa_planes_counters.txt 2>&1
tail -24 gpurun_out/r02_pmc_ipa_planes_counters.txt
