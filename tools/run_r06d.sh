# round 6: the pair stream non-temporal -- end to end.  tree = edge transition nt (loads + stores), edge embedding plain;
# etplain = edge transition with plain accesses (the kernel of round 5); eent = tree + nt stores in the edge embedding.  Interleaved, one call.
O=gpurun_out/r06d; mkdir -p $O; D=str2str_amd/csrc/build
run() { n=$1; cfg=$2; lib=$3; if [ $lib = tree ]; then P=""; else P="STR2STR_HIP_LIB=$PWD/$D/ab_$lib.so"; fi
  env $P python bench.py --config $cfg --steps 1 --warmup 1 --no-cpu-baseline --no-kernel-table --no-other-configs > $O/$n.json 2> $O/$n.err; }
for rep in a b; do
  for cfg in cfg2 cfg3 ref_default; do
    for lib in tree etplain eent; do run ${cfg}_${lib}_$rep $cfg $lib; done
  done
done
python - <<PY
import json, glob, os
for f in sorted(glob.glob("$O/*_[ab].json")):
    try:
        l = json.loads(open(f).read().strip().splitlines()[-1]); print(os.path.basename(f)[:-5], round(l["value"], 2), round(l["ms_per_step"], 1))
    except Exception as e:
        print(os.path.basename(f), "ERR", e); print(open(f[:-4] + "err").read()[-600:])
PY
