#!/bin/bash
# every csrc/*_kernels.hip with the product flags + extra flags -> scripts/diag/_bin/libias_<name>.so
#   bash scripts/diag/build_allflags.sh <name> <extra flags...>
R=$(cd $(dirname $0)/../.. && pwd); C=$R/inverse-audio-synthesis_amd/csrc; name=$1; shift
B=$R/scripts/diag/_bin/all_$name; mkdir -p $B
for f in $C/*_kernels.hip; do
  s=$(basename $f .hip); extra=""
  case $s in voice_kernels|voice_grad_kernels) extra="-ffp-contract=off";; esac
  /opt/rocm/bin/hipcc -O3 -fPIC -std=c++17 --offload-arch=gfx950 -Wno-unused-value -mllvm -amdgpu-kernarg-preload-count=16 -fno-slp-vectorize $extra "$@" -c $f -o $B/$s.o &
  while [ $(jobs -r | wc -l) -ge 4 ]; do sleep 1; done
done
wait
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 $B/*.o -o $R/scripts/diag/_bin/libias_$name.so && echo built $name
