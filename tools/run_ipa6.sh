mkdir -p gpurun_out/ipa6
export GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$PWD}
timeout 1200 python -m pytest tests -m gpu -x -q -k "ipa_vs_oracle or ipa_golden" 2>&1 | tail -4
bash tools/pmc_ipa_planes.sh gpurun_out/r02_pmc_ipa_planes_traffic.json 128 256 > gpurun_out/ipa6/pmc.log 2>&1; tail -32 gpurun_out/ipa6/pmc.log
