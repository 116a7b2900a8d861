#!/bin/bash
# scripts/dev/mkvariant.sh <name> <extra flags...> : variant lib scratch/ab/lib_<name>.so with part 1 of the pure unit (pressure-only
# VLE kernel, Jacobians, C ABI: the flags of feos_torch_amd/build.py) recompiled with the extra flags; everything else from build/
set -e
NAME=$1; shift
C=feos_torch_amd/csrc; B=feos_torch_amd/build
mkdir -p scratch/ab
hipcc -O3 --offload-arch=gfx950 -std=c++17 -fPIC -fno-honor-nans -fno-honor-infinities -fno-signed-zeros -fno-slp-vectorize -fassociative-math -freciprocal-math -DPCS_FAST_RCP -DPCS_FAST_LOG -DPCS_F32_PRESOLVE -DPCS_PURE_PART=1 "$@" -c -o scratch/ab/pure_$NAME.o $C/pure_kernels.hip
hipcc --offload-arch=gfx950 -shared -fPIC -o scratch/ab/lib_$NAME.so scratch/ab/pure_$NAME.o $B/pure_kernels_b.o $B/pure_robust.o $B/compact_kernels.o $B/mix_kernels.o $B/mixn_kernels.o $B/gc_kernels.o $B/gc_gradient.o
rm scratch/ab/pure_$NAME.o
