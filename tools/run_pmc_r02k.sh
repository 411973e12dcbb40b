#!/bin/bash
# end-of-round-2 PMC evidence for the f16x3 kernels: issue / stall / MFMA-busy counter groups (one group per run, kernel-trace only)
export GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$PWD}
mkdir -p gpurun_out
bash tools/pmc_kernel.sh r02k_et edge_transition_f16 -- python tools/et_only.py --B 128 --N 256 --iters 4 --proj --mode f16x3 > gpurun_out/r02k_pmc_et_f16_counters.txt 2>&1
bash tools/pmc_kernel.sh r02k_ipa ipa_attention_f16 -- python tools/ipa_loop.py --seconds 1 --path f16 > gpurun_out/r02k_pmc_ipa_f16_counters.txt 2>&1
bash tools/pmc_kernel.sh r02k_ee edge_embed_f16 -- python tools/ee_time.py > gpurun_out/r02k_pmc_ee_f16_counters.txt 2>&1
rm -rf gpurun_out/pmc_r02k_et gpurun_out/pmc_r02k_ipa gpurun_out/pmc_r02k_ee
tail -22 gpurun_out/r02k_pmc_et_f16_counters.txt; tail -22 gpurun_out/r02k_pmc_ipa_f16_counters.txt | head -30
