# round 6: GPU suite, the encoder layer's post-attention half as one launch (S2S_ENC_CHAIN3) A/B on cfg2 / cfg3 / ref_default in ONE call,
# rocprofv3 kernel stats + the launch sequence of one evaluation
T=r06b; O=gpurun_out/$T; mkdir -p $O
python -m pytest tests -m gpu -q -x 2>&1 | tail -4 > $O/pytest.log; cat $O/pytest.log
run() { n=$1; cfg=$2; shift 2; env "$@" python bench.py --config $cfg --steps 1 --warmup 1 --no-cpu-baseline --no-kernel-table --no-other-configs > $O/$n.json 2> $O/$n.err; }
for rep in a b; do
  run cfg2_chain3_$rep cfg2 S2S_ENC_CHAIN3=1
  run cfg2_sep_$rep cfg2 S2S_ENC_CHAIN3=0
  run cfg3_chain3_$rep cfg3 S2S_ENC_CHAIN3=1
  run cfg3_sep_$rep cfg3 S2S_ENC_CHAIN3=0
  run ref_chain3_$rep ref_default S2S_ENC_CHAIN3=1
  run ref_sep_$rep ref_default S2S_ENC_CHAIN3=0
done
python - <<PY
import json, glob, os
for f in sorted(glob.glob("$O/*_[ab].json")):
    try:
        l = json.loads(open(f).read().strip().splitlines()[-1]); print(os.path.basename(f), round(l["value"], 2), round(l["ms_per_step"], 1))
    except Exception as e:
        print(os.path.basename(f), "ERR", e); print(open(f[:-4] + "err").read()[-800:])
PY
tools/prof_bench.sh $T
DB=$(ls gpurun_out/prof_$T/*/*results.db gpurun_out/prof_$T/*results.db 2>/dev/null | head -1)
python tools/rocpd_sequence.py $DB gpurun_out/${T}_eval_sequence.md | tail -3
rm -rf gpurun_out/prof_$T
