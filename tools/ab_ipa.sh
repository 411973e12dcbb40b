# same-call A/B of one IPA block (folded operands) across library builds:  bash tools/ab_ipa.sh <lib.so> <lib.so> ...   ("tree" = the library in the tree)
for rep in 1 2; do
for L in "$@"; do
  for shape in "128 256" "1000 35" "1000 80" "32 512"; do
    bb=${shape% *}; nn=${shape#* }
    if [ $L = tree ]; then P=""; else P="STR2STR_HIP_LIB=$PWD/$L"; fi
    echo -n "$L: "; env $P python tools/ipa_fold_ab.py --B $bb --N $nn --iters 10 --only-folded 2>/dev/null | tail -2 | tr '\n' ' '; echo
  done
done
done
