#!/bin/bash
# one translation unit rebuilt with extra flags and linked with the other objects of the tree into variants/libNAME.so
# usage: bash scripts/mkvariant.sh NAME FILE.hip "EXTRA FLAGS"     (run after `make` in art_amd/csrc)
set -e
cd "$(dirname "$0")/../art_amd/csrc"
N=$1; F=$2; X=$3
mkdir -p ../../variants
B=${F%.hip}
EXTRA=""
case $B in
  nlm_sweep) EXTRA="-fgpu-flush-denormals-to-zero -fno-slp-vectorize";;
  amaze_stream) EXTRA="-mllvm -amdgpu-sched-strategy=max-ilp";;
  nlmeans) EXTRA="-fno-slp-vectorize";;
  shrinkblur) EXTRA="-mllvm -amdgpu-sched-strategy=max-memory-clause";;
esac
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wno-unused-function $EXTRA $X -c $F -o /tmp/var_$N.o
OBJS=$(ls *.o | grep -v "^$B.o$")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../variants/lib$N.so $OBJS /tmp/var_$N.o
echo variants/lib$N.so
