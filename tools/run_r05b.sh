O=gpurun_out/r05b; mkdir -p $O
python -m pytest tests -m gpu -q -x -k "merged or predict_step or eval_entry or se3 or embed or multirank or smoke" 2>&1 | tail -15 > $O/pytest_sel.log
cat $O/pytest_sel.log
S2S_MERGE_DELTAS=0 python bench.py --config ref_default --steps 1 --warmup 0 > $O/bench_ref_default_off.json 2> $O/off.err
python bench.py --config ref_default --steps 1 --warmup 0 > $O/bench_ref_default_on.json 2> $O/on.err
python bench.py --config ref_default --steps 1 --warmup 0 --rng host > $O/bench_ref_default_on_host.json 2> $O/onh.err
python - <<PY
import json
for n in ("off","on","on_host"):
    try:
        l=json.loads(open("$O/bench_ref_default_%s.json"%n).read().strip().splitlines()[-1]); print(n, round(l["value"],2), round(l["ms_per_step"]))
    except Exception as e: print(n, "ERR", e); print(open("$O/%s.err"%{"off":"off","on":"on","on_host":"onh"}[n]).read()[-1500:])
PY
