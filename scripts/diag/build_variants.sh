#!/bin/bash
# Variant builds of one csrc/*_kernels.hip for A/B runs through IAS_HIP_LIB (developer tool):
#   bash scripts/diag/build_variants.sh <file stem, e.g. pqmf_kernels> "<name> <flags>" ...  -> scripts/diag/_bin/libias_<name>.so
R=$(cd $(dirname $0)/../.. && pwd)
C=$R/inverse-audio-synthesis_amd/csrc
B=$R/scripts/diag/_bin
mkdir -p $B
stem=$1; shift
objs=$(ls $C/*_kernels.o | grep -v $stem.o)
extra="-fno-slp-vectorize"       # the product flags (csrc/Makefile)
[ $stem = voice_kernels ] && extra="-ffp-contract=off -fno-slp-vectorize"
[ $stem = voice_grad_kernels ] && extra="-ffp-contract=off -fno-slp-vectorize"
for spec in "$@"; do
  set -- $spec; name=$1; shift
  /opt/rocm/bin/hipcc -O3 -fPIC -std=c++17 --offload-arch=gfx950 -Wno-unused-value -mllvm -amdgpu-kernarg-preload-count=16 $extra "$@" -c $C/$stem.hip -o $B/${stem}_$name.o || exit 1
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 $B/${stem}_$name.o $objs -o $B/libias_$name.so
  echo built $name
done
