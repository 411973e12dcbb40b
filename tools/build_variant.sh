#!/bin/bash
# tools/build_variant.sh <name> [extra hipcc flags]: library variant with one differently compiled unit
# (UNIT=pair_mlp_f16 by default = the edge transition for chains of 32+ residues; UNIT=pair_mlp_f16_b its short-chain form, UNIT=pair_mlp_f16_c the
# edge embedding -- three translation units of one source, with SRC=<file> for _b / _c a wrapper that includes the other version; e.g. UNIT=ipa_attention tools/build_variant.sh qr0 -DS2S_IPA_QR=0).  Run
# `python -m str2str_amd.build` first: the other units are linked from its objects.
# NOTE: .gpurunignore lists str2str_amd/csrc/build/ab_*.so (45 stale variants once cost every lease a 73 MB push): comment that line out for an
# A/B session and delete the variants afterwards (rm str2str_amd/csrc/build/ab_*.so).
N=$1; shift
U=${UNIT:-pair_mlp_f16}
D=str2str_amd/csrc/build
# the unit's own flags come from str2str_amd/build.py (e.g. -fno-slp-vectorize of the MFMA units: a variant built without them is not comparable)
FLAGS=$(python -c "from str2str_amd.build import UNITS, COMMON; print(' '.join(COMMON + UNITS['$U.hip']))")
SRC=${SRC:-str2str_amd/csrc/$U.hip}     # SRC=<file>: another version of the unit's source (e.g. from git show)
hipcc -x hip -c $SRC -o $D/${U}_$N.o $FLAGS -w "$@" || exit 1
OBJS=""
for u in $(python -c "from str2str_amd.build import UNITS; print(' '.join(k.rsplit('.',1)[0] for k in UNITS))"); do
  if [ $u == $U ]; then OBJS="$OBJS $D/${U}_$N.o"; else OBJS="$OBJS $D/$u.o"; fi
done
hipcc -shared -fPIC --offload-arch=gfx950 -o $D/ab_$N.so $OBJS && echo $D/ab_$N.so
