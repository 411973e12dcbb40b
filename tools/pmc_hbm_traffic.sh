#!/bin/bash
# HBM traffic per kernel from PMC counters, collected as /opt/skills/guides/MI355X_MICROARCH.md prescribes:
# FETCH_SIZE and WRITE_SIZE in SEPARATE rocprofv3 --pmc passes (kernel-trace only, no other trace domains); both count
# KiB per dispatch; on gfx950 FETCH_SIZE under-reports wide coalesced reads by 2x, so the read side is doubled.
#   tools/pmc_hbm_traffic.sh <out.json> [B] [N]
OUTJSON=$1; B=${2:-16}; N=${3:-256}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/pmc_traffic; rm -rf $OUT; mkdir -p $OUT
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/$c -- python tools/kernel_bench.py --B $B --N $N --iters 2 > $OUT/$c.log 2>&1
done
python - <<PY
import csv, glob, json, collections
B, N = $B, $N
pairs = B * N * N
names = {"edge_transition_f16": ("edge_transition_f16x3", (2 * (1024 + 160) + 512 + 160) / 3),   # the trunk's three launches (tools/kernel_bench.py) "edge_transition_kernel": ("edge_transition", 1024),
         "edge_embed_f16_kernel": ("edge_embed_f16x3", 512 + 160), "edge_embed_kernel": ("edge_embed", 512 + 160),
         "pair_project_kernel": ("pair_project", 672), "ipa_attention_f16w_kernel": ("ipa_attention", None),
         "ipa_opair_kernel": ("ipa_opair", None)}
acc = {v[0]: {"FETCH_SIZE": [], "WRITE_SIZE": []} for v in names.values()}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob("$OUT/%s/**/*counter_collection.csv" % c, recursive=True):
        for r in csv.DictReader(open(f)):
            for k, (nm, _) in names.items():
                if k in r["Kernel_Name"] and r["Counter_Name"] == c:
                    acc[nm][c].append(float(r["Counter_Value"]) * 1024.0)
out = {"source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) on tools/kernel_bench.py "
                 "--B %d --N %d; counters are KiB per dispatch; gfx950 FETCH_SIZE under-reports wide coalesced reads by 2x "
                 "(MI355X_MICROARCH.md, HBM section): 'hbm_bytes_corrected' doubles the read side" % (B, N),
       "shape": {"B": B, "N": N, "pairs": pairs}, "kernels": {}}
ipa_alg = B * 4 * (9512 * N + 40 * N * N)
for nm, d in acc.items():
    if not d["FETCH_SIZE"] or not d["WRITE_SIZE"]:
        continue
    fe, wr = sum(d["FETCH_SIZE"]) / len(d["FETCH_SIZE"]), sum(d["WRITE_SIZE"]) / len(d["WRITE_SIZE"])
    e = {"fetch_bytes": fe, "write_bytes": wr, "hbm_bytes_raw": fe + wr, "hbm_bytes_corrected": 2 * fe + wr,
         "bytes_per_pair_corrected": (2 * fe + wr) / pairs}
    alg = [v[1] for v in names.values() if v[0] == nm][0]
    if alg:
        e["algorithmic_bytes_per_pair"] = alg
    out["kernels"][nm] = e
if "ipa_attention" in out["kernels"] and "ipa_opair" in out["kernels"]:
    tot = out["kernels"]["ipa_attention"]["hbm_bytes_corrected"] + out["kernels"]["ipa_opair"]["hbm_bytes_corrected"]
    out["ipa_attention_plus_opair"] = {"hbm_bytes_corrected": tot, "algorithmic_bytes": ipa_alg, "ratio": tot / ipa_alg}
json.dump(out, open("$OUTJSON", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
