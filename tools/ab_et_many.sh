# same-call A/B of the edge transition at cfg2's call: the tree's library against several variants, interleaved (3 rounds)
#   bash tools/ab_et_many.sh <a.so> <b.so> ...      (variants from tools/build_variant.sh: built with the unit's flags of build.py)
run() { STR2STR_HIP_LIB=$1 python tools/et_only.py --B 128 --N 256 --iters 20 --proj --layout tiled 2>/dev/null | tail -1 | sed "s|^|$2: |"; }
for rep in 1 2 3; do
  run "" tree
  for v in "$@"; do run $PWD/$v $(basename $v .so); done
done
