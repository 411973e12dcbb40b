#!/bin/bash
# HBM traffic of the IPA attention kernels (planes path and fp32-operand path, + s2s_ipa_opair) from PMC counters, collected as
# /opt/skills/guides/MI355X_MICROARCH.md prescribes (FETCH_SIZE / WRITE_SIZE in separate --pmc passes, kernel-trace only; KiB per
# dispatch; gfx950 FETCH_SIZE under-reports wide coalesced reads by 2x, so the read side is doubled).
#   tools/pmc_ipa_planes.sh <out.json> [B] [N]
OUTJSON=$1; B=${2:-128}; N=${3:-256}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/pmc_ipa_planes; rm -rf $OUT; mkdir -p $OUT
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 900 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/$c -- python tools/ipa_block_bench.py --B $B --N $N --iters 2 > $OUT/$c.log 2>&1
done
python - <<PY
import csv, glob, json
B, N = $B, $N
names = {"ipa_attention_f16w_kernel": "ipa_attention_f16w", "ipa_attention_f16_kernel": "ipa_attention_f16", "ipa_attention_planes_kernel": "ipa_attention_planes", "ipa_attention_kernel": "ipa_attention_fp32_operands", "ipa_opair_kernel": "ipa_opair",
         "ipa_prep_planes_kernel": "ipa_prep_points_planes"}
acc = {v: {"FETCH_SIZE": [], "WRITE_SIZE": []} for v in names.values()}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob("$OUT/%s/**/*counter_collection.csv" % c, recursive=True):
        for r in csv.DictReader(open(f)):
            for k, nm in names.items():
                if k in r["Kernel_Name"] and r["Counter_Name"] == c:
                    acc[nm][c].append(float(r["Counter_Value"]) * 1024.0)
alg = B * 4 * (9512 * N + 40 * N * N)
out = {"source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) on tools/ipa_block_bench.py --B %d --N %d; "
                 "KiB per dispatch; read side doubled (gfx950 FETCH_SIZE correction, MI355X_MICROARCH.md HBM section)" % (B, N),
       "shape": {"B": B, "N": N}, "algorithmic_bytes_attention_plus_opair": alg, "kernels": {}}
for nm, d in acc.items():
    if d["FETCH_SIZE"] and d["WRITE_SIZE"]:
        fe, wr = sum(d["FETCH_SIZE"]) / len(d["FETCH_SIZE"]), sum(d["WRITE_SIZE"]) / len(d["WRITE_SIZE"])
        out["kernels"][nm] = {"fetch_bytes": fe, "write_bytes": wr, "hbm_bytes_corrected": 2 * fe + wr}
k = out["kernels"]
for path, att in (("f16w", "ipa_attention_f16w"), ("f16", "ipa_attention_f16"), ("planes", "ipa_attention_planes"), ("fp32_operands", "ipa_attention_fp32_operands")):
    if att in k and "ipa_opair" in k:
        tot = k[att]["hbm_bytes_corrected"] + k["ipa_opair"]["hbm_bytes_corrected"]
        out["attention_plus_opair_" + path] = {"hbm_bytes_corrected": tot, "ratio_to_algorithmic": tot / alg}
json.dump(out, open("$OUTJSON", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
