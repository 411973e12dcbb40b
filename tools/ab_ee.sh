# same-call A/B of the edge embedding (cfg2 shape, fused projection, tiled output): the tree against another build
#   bash tools/ab_ee.sh <other.so>
for rep in 1 2 3; do
EE_LAYOUT=tiled EE_ITERS=20 python tools/ee_time.py 128 256 2>/dev/null | tail -1
STR2STR_HIP_LIB=$PWD/$1 EE_LAYOUT=tiled EE_ITERS=20 python tools/ee_time.py 128 256 2>/dev/null | tail -1
done
