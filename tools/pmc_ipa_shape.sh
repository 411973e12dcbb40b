#!/bin/bash
# HBM traffic of the IPA launch pair (attention + pair term) at ANY shape, folded operands as the network runs them -- the ragged / short-chain
# instances cfg3 launches:   tools/pmc_ipa_shape.sh <out.json> <B> <N>
# Same recipe as tools/pmc_ipa.sh (FETCH_SIZE and WRITE_SIZE in SEPARATE rocprofv3 --pmc passes, kernel trace only; KiB per dispatch;
# gfx950 FETCH_SIZE x 2 for wide coalesced reads), on tools/ipa_fold_ab.py --only-folded.
OUTJSON=$1; B=$2; N=$3
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/pmc_ipa_shape; rm -rf $OUT; mkdir -p $OUT
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/$c -- python tools/ipa_fold_ab.py --B $B --N $N --iters 2 --only-folded > $OUT/$c.log 2>&1
done
python - <<PY
import csv, glob, json
B, N = $B, $N
names = {"ipa_attention": "ipa_attention", "ipa_opair_kernel": "ipa_opair"}
acc = {v: {"FETCH_SIZE": [], "WRITE_SIZE": [], "kernel_names": set()} for v in names.values()}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob("$OUT/%s/**/*counter_collection.csv" % c, recursive=True):
        for r in csv.DictReader(open(f)):
            for k, nm in names.items():
                if k in r["Kernel_Name"] and r["Counter_Name"] == c:
                    acc[nm][c].append(float(r["Counter_Value"]) * 1024.0)
                    acc[nm]["kernel_names"].add(r["Kernel_Name"][:80])
alg = B * 4 * (9512 * N + 40 * N * N)
out = {"source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) on tools/ipa_fold_ab.py --only-folded (B = %d, N = %d); "
                 "KiB per dispatch; gfx950 FETCH_SIZE under-reports wide coalesced reads by 2x: 'hbm_bytes_corrected' doubles the read side" % (B, N),
       "algorithmic_bytes_per_launch_pair": alg, "kernels": {}}
tot = 0.0
for nm, d in acc.items():
    if d["FETCH_SIZE"] and d["WRITE_SIZE"]:
        fe, wr = sum(d["FETCH_SIZE"]) / len(d["FETCH_SIZE"]), sum(d["WRITE_SIZE"]) / len(d["WRITE_SIZE"])
        out["kernels"][nm] = {"fetch_bytes": fe, "write_bytes": wr, "hbm_bytes_corrected": 2 * fe + wr, "hbm_bytes_raw": fe + wr,
                              "dispatches": len(d["FETCH_SIZE"]), "kernel_names": sorted(d["kernel_names"])}
        tot += 2 * fe + wr
out["attention_plus_opair"] = {"hbm_bytes_corrected": tot, "ratio_to_algorithmic": tot / alg}
json.dump(out, open("$OUTJSON", "w"), indent=1)
print(json.dumps(out["attention_plus_opair"]))
PY
rm -rf $OUT
