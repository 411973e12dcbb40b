# host-noise (parity mode) default block: host start frames on one intra-op thread vs the whole pool, fast-forward vs real draws; GPU suite
O=gpurun_out/r05k; mkdir -p $O
python -m pytest tests -m gpu -q -x 2>&1 | tail -4 > $O/pytest.log; cat $O/pytest.log
run() { n=$1; shift; env "$@" python bench.py --config ref_default --steps 1 --warmup 0 --rng host > $O/$n.json 2> $O/$n.err; }
run pool S2S_HOST_FM_THREADS=0
run one S2S_HOST_FM_THREADS=1
run one_b S2S_HOST_FM_THREADS=1
run pool_b S2S_HOST_FM_THREADS=0
run draws S2S_HOST_RNG_FAST=0
python bench.py --config ref_default --steps 1 --warmup 0 > $O/device.json 2> $O/device.err
python - <<PY
import json
for n in ("pool","one","one_b","pool_b","draws","device"):
    try:
        l=json.loads(open("$O/%s.json"%n).read().strip().splitlines()[-1]); print(n, round(l["value"],2), round(l["ms_per_step"]))
    except Exception as e: print(n, "ERR", e); print(open("$O/%s.err"%n).read()[-1500:])
PY
