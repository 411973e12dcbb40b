#!/bin/bash
# HBM traffic of the IPA launch pair (s2s_ipa_attention_f16w + s2s_ipa_opair) from PMC counters, per the guide's recipe (FETCH_SIZE and
# WRITE_SIZE in SEPARATE rocprofv3 --pmc passes, kernel trace only; KiB per dispatch; gfx950 FETCH_SIZE x 2 for wide coalesced reads).
#   tools/pmc_ipa.sh <out.json>          (cfg2 shape: B = 128, N = 256, tools/ipa_loop.py)
OUTJSON=$1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/pmc_ipa; rm -rf $OUT; mkdir -p $OUT
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/$c -- python tools/ipa_loop.py --seconds 0.5 > $OUT/$c.log 2>&1
done
python - <<PY
import csv, glob, json
B, N = 128, 256
names = {"ipa_attention_f16w_kernel": "ipa_attention_f16w", "ipa_opair_kernel": "ipa_opair"}
acc = {v: {"FETCH_SIZE": [], "WRITE_SIZE": []} for v in names.values()}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob("$OUT/%s/**/*counter_collection.csv" % c, recursive=True):
        for r in csv.DictReader(open(f)):
            for k, nm in names.items():
                if k in r["Kernel_Name"] and r["Counter_Name"] == c:
                    acc[nm][c].append(float(r["Counter_Value"]) * 1024.0)
alg = B * 4 * (9512 * N + 40 * N * N)
out = {"source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) on tools/ipa_loop.py (B = 128, N = 256); KiB per "
                 "dispatch; gfx950 FETCH_SIZE under-reports wide coalesced reads by 2x: 'hbm_bytes_corrected' doubles the read side",
       "algorithmic_bytes_per_launch_pair": alg, "kernels": {}}
tot = 0.0
for nm, d in acc.items():
    if d["FETCH_SIZE"] and d["WRITE_SIZE"]:
        fe, wr = sum(d["FETCH_SIZE"]) / len(d["FETCH_SIZE"]), sum(d["WRITE_SIZE"]) / len(d["WRITE_SIZE"])
        out["kernels"][nm] = {"fetch_bytes": fe, "write_bytes": wr, "hbm_bytes_corrected": 2 * fe + wr, "dispatches": len(d["FETCH_SIZE"])}
        tot += 2 * fe + wr
out["attention_plus_opair"] = {"hbm_bytes_corrected": tot, "ratio_to_algorithmic": tot / alg}
json.dump(out, open("$OUTJSON", "w"), indent=1)
print(json.dumps(out["attention_plus_opair"]))
PY
rm -rf $OUT
