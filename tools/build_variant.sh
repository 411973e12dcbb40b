#!/bin/bash
# tools/build_variant.sh <name> [extra hipcc flags]: library variant with a differently compiled pair_mlp_bf16 unit
N=$1; shift
D=str2str_amd/csrc/build
hipcc -x hip -c str2str_amd/csrc/pair_mlp_bf16.hip -o $D/pair_mlp_bf16_$N.o -O3 -std=c++17 -fPIC --offload-arch=gfx950 -I include -I str2str_amd/csrc -mllvm -pragma-unroll-threshold=10000000 "$@" || exit 1
hipcc -shared -fPIC --offload-arch=gfx950 -o $D/lib_$N.so $D/abi.o $D/rigid_kernels.o $D/se3_step.o $D/pair_mlp.o $D/pair_mlp_bf16_$N.o $D/ipa_attention.o && echo $D/lib_$N.so
