# same-call A/B of the edge embedding (cfg2 shape, fused projection, tiled output): the tree against other builds, interleaved
#   bash tools/ab_ee.sh <a.so> [<b.so> ...]      (variants: UNIT=pair_mlp_f16_c bash tools/build_variant.sh <name> [-D...], built with the unit's flags of build.py)
for rep in 1 2 3; do
EE_LAYOUT=tiled EE_ITERS=20 python tools/ee_time.py 128 256 2>/dev/null | tail -1
for v in "$@"; do STR2STR_HIP_LIB=$PWD/$v EE_LAYOUT=tiled EE_ITERS=20 python tools/ee_time.py 128 256 2>/dev/null | tail -1; done
done
