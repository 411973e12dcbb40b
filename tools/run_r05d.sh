# GPU tests on the tree + PMC passes of round 5's edge transition (HBM traffic at B = 16; issue / stall / MFMA-busy counter groups)
O=gpurun_out/r05d; mkdir -p $O
python -m pytest tests -m gpu -q -x 2>&1 | tail -4 > $O/pytest.log; cat $O/pytest.log
bash tools/pmc_hbm_traffic.sh gpurun_out/r05d_pmc_hbm_traffic.json 16 256 > $O/pmc_traffic.log 2>&1; tail -3 $O/pmc_traffic.log
bash tools/pmc_kernel.sh r05d_et edge_transition_f16 -- python tools/et_only.py --B 64 --N 256 --iters 2 --proj --layout tiled > gpurun_out/r05d_pmc_et_f16_counters.txt 2>&1
tail -24 gpurun_out/r05d_pmc_et_f16_counters.txt
rm -rf gpurun_out/pmc_r05d_et gpurun_out/pmc_traffic
