#!/bin/bash
# round-2 PMC evidence: (1) issue / stall / MFMA-busy counters of the dominant kernel at the cfg2 shape (fused projection variant),
# (2) HBM traffic passes (FETCH_SIZE / WRITE_SIZE, separate runs) at B = 16, N = 256
export GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$PWD}
bash tools/pmc_kernel.sh r02_et edge_transition_bf16 -- python tools/et_only.py --B 128 --N 256 --iters 4 --proj > gpurun_out/r02_pmc_et_counters.txt 2>&1
bash tools/pmc_kernel.sh r02_ipa ipa_attention_kernel -- python tools/ipa_only.py --B 128 --N 256 --iters 4 > gpurun_out/r02_pmc_ipa_counters.txt 2>&1
bash tools/pmc_hbm_traffic.sh gpurun_out/r02_pmc_hbm_traffic.json 16 256 > gpurun_out/r02_pmc_hbm_traffic.log 2>&1
tail -30 gpurun_out/r02_pmc_et_counters.txt; tail -25 gpurun_out/r02_pmc_ipa_counters.txt; tail -5 gpurun_out/r02_pmc_hbm_traffic.log
