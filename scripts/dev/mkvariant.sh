#!/bin/bash
# scratch/mkvariant.sh <name> <extra -D flags...> : build a variant lib scratch/ab/lib_<name>.so (pure TU rebuilt with the flags)
set -e
NAME=$1; shift
C=feos_torch_amd/csrc; B=feos_torch_amd/build
mkdir -p scratch/ab
hipcc -O3 --offload-arch=gfx950 -std=c++17 -fPIC -fno-honor-nans -fno-honor-infinities -fno-signed-zeros -fno-slp-vectorize -DPCS_FAST_RCP -DPCS_F32_PRESOLVE "$@" -c -o scratch/ab/pure_$NAME.o $C/pure_kernels.hip
hipcc --offload-arch=gfx950 -shared -fPIC -o scratch/ab/lib_$NAME.so scratch/ab/pure_$NAME.o $B/pure_robust.o $B/compact_kernels.o $B/mix_kernels.o $B/mixn_kernels.o $B/gc_kernels.o $B/gc_gradient.o
rm scratch/ab/pure_$NAME.o
