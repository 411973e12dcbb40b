timeout 900 python -m pytest tests -m gpu -x -q -k "edge_transition" 2>&1 | tail -5
for m in bf16x6 f16x3 bf16x6 f16x3; do S2S_EDGE_MFMA=$m python tools/et_only.py --B 128 --N 256 --iters 20 --proj --mode $m 2>/dev/null | tail -1; done
