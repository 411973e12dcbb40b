export STR2STR_HIP_LIB=$PWD/str2str_amd/csrc/build/lib_ee2.so
python tools/ee_time.py 2>/dev/null
timeout 900 python -m pytest tests -m gpu -x -q -k "embed or forward or trajectory or teacher" 2>&1 | tail -5
