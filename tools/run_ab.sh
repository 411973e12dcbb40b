echo "== base"; for i in 1 2; do python tools/et_only.py --B 128 --N 256 --iters 20 --proj 2>/dev/null | tail -1; done
echo "== rolled epilogues"; for i in 1 2; do STR2STR_HIP_LIB=$PWD/str2str_amd/csrc/build/lib_etroll.so python tools/et_only.py --B 128 --N 256 --iters 20 --proj 2>/dev/null | tail -1; done
echo "== base again"; python tools/et_only.py --B 128 --N 256 --iters 20 --proj 2>/dev/null | tail -1
STR2STR_HIP_LIB=$PWD/str2str_amd/csrc/build/lib_etroll.so timeout 900 python -m pytest tests -m gpu -x -q -k "edge_transition or net_golden or free_running or many_tiles or float64" 2>&1 | tail -3
