#!/bin/bash
# scripts/dev/mkgc.sh <name> <flags...>: variant lib scratch/ab/lib_<name>.so with gc_kernels.hip rebuilt
set -e
NAME=$1; shift
C=feos_torch_amd/csrc; B=feos_torch_amd/build
mkdir -p scratch/ab
hipcc -O3 --offload-arch=gfx950 -std=c++17 -fPIC "$@" -c -o scratch/ab/gc_$NAME.o $C/gc_kernels.hip
hipcc --offload-arch=gfx950 -shared -fPIC -o scratch/ab/lib_$NAME.so $B/pure_kernels.o $B/pure_kernels_b.o $B/pure_robust.o $B/compact_kernels.o $B/mix_kernels.o $B/mixn_kernels.o scratch/ab/gc_$NAME.o $B/gc_gradient.o
rm scratch/ab/gc_$NAME.o
