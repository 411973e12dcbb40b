# same-call A/B of the edge transition: the library in the tree against a variant / an older build (STR2STR_HIP_LIB)
#   bash tools/ab_et_barrier.sh <other.so>
for rep in 1 2 3; do
python tools/et_only.py --B 128 --N 256 --iters 20 --proj --layout tiled 2>/dev/null | tail -1
STR2STR_HIP_LIB=$PWD/$1 python tools/et_only.py --B 128 --N 256 --iters 20 --proj --layout tiled 2>/dev/null | tail -1
done
python tools/et_only.py --B 100 --N 35 --iters 50 --proj --layout tiled 2>/dev/null | tail -1
STR2STR_HIP_LIB=$PWD/$1 python tools/et_only.py --B 100 --N 35 --iters 50 --proj --layout tiled 2>/dev/null | tail -1
