for rep in 1 2 3; do
  for v in ph16 ph1; do
    L=str2str_amd/csrc/build/ab_$v.so
    for shape in "128 256" "100 35" "100 80" "1000 35" "1000 10"; do
      set -- $shape
      echo -n "$v B=$1 N=$2: "; STR2STR_HIP_LIB=$L python tools/et_only.py --B $1 --N $2 --proj --layout tiled --iters 20 2>/dev/null | sed 's/.*: //'
    done
  done
done
