mkdir -p gpurun_out
python tools/ee_probe.py run --B 128 --N 256 > gpurun_out/ee_probe.txt 2>&1
for v in "" ee4 "" ee4; do
if [ -n "$v" ]; then export STR2STR_HIP_LIB=$PWD/str2str_amd/csrc/build/lib_$v.so; else unset STR2STR_HIP_LIB; fi
python tools/ee_time.py 2>/dev/null
done
timeout 900 python -m pytest tests -m gpu -x -q -k "embed or forward or trajectory or teacher" 2>&1 | tail -3
