#!/bin/bash
# scratch/mkmix.sh <name> <flags...>: variant lib with mix_kernels.hip rebuilt
set -e
mkdir -p scratch/ab
NAME=$1; shift
C=feos_torch_amd/csrc; B=feos_torch_amd/build
hipcc -O3 --offload-arch=gfx950 -std=c++17 -fPIC "$@" -c -o scratch/ab/mix_$NAME.o $C/mix_kernels.hip
hipcc --offload-arch=gfx950 -shared -fPIC -o scratch/ab/lib_$NAME.so $B/pure_kernels.o $B/pure_kernels_b.o $B/pure_robust.o $B/compact_kernels.o scratch/ab/mix_$NAME.o $B/mixn_kernels.o $B/gc_kernels.o $B/gc_gradient.o
rm scratch/ab/mix_$NAME.o
