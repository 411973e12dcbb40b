# evidence run of round 5 (one gpurun call): box calibration, GPU tests, smoke, the default bench line (cfg2 + other_configs)
#   bash tools/run_r05.sh <tag> [--prof]     --prof: also rocprofv3 kernel stats + launch sequence of one cfg2 step
T=${1:-r05a}
O=gpurun_out/$T; mkdir -p $O
python tools/et_only.py --B 128 --N 256 --iters 20 --proj --layout tiled 2>/dev/null | tail -1 > $O/box_calibration.txt
python -m pytest tests -m gpu -q -x 2>&1 | tail -6 > $O/pytest.log
python __graft_entry__.py --smoke 2>&1 | tail -1 > $O/smoke.txt
python bench.py > $O/bench_cfg2.json 2> $O/bench_cfg2.err
cat $O/box_calibration.txt $O/pytest.log $O/smoke.txt
python - <<PY
import json
l=json.loads(open("$O/bench_cfg2.json").read().strip().splitlines()[-1])
print("other_configs", {k: round(v.get("value", 0), 2) for k, v in l.get("other_configs", {}).items()}); print("cfg2", round(l["value"],3), "conf/s", round(l["ms_per_step"],1), "ms/step", "roofline", round(l.get("roofline",{}).get("frac") or 0,4), "ipa", round(l.get("ipa_kernel",{}).get("frac") or 0,4))
print({k: (round(v["total_ms"],1), v["launches"]) for k, v in l.get("kernel_times", {}).items() if isinstance(v, dict)})
PY
if [ "$2" = "--prof" ]; then
  bash tools/prof_bench.sh $T > $O/prof.log 2>&1
  head -14 gpurun_out/${T}_bench_kernel_stats.md
  DB=$(ls gpurun_out/prof_$T/*/*results.db gpurun_out/prof_$T/*results.db 2>/dev/null | head -1)
  python tools/rocpd_sequence.py $DB gpurun_out/${T}_eval_sequence.md > /dev/null
  rm -rf gpurun_out/prof_$T
fi
