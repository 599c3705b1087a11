#!/bin/bash
# every csrc/*_kernels.hip with -fno-slp-vectorize -> scripts/diag/_bin/libias_allnoslp.so (A/B against the product flags)
R=$(cd $(dirname $0)/../.. && pwd); C=$R/inverse-audio-synthesis_amd/csrc; B=$R/scripts/diag/_bin/allnoslp; mkdir -p $B
for f in $C/*_kernels.hip; do
  s=$(basename $f .hip); extra=""
  case $s in voice_kernels|voice_grad_kernels) extra="-ffp-contract=off";; esac
  /opt/rocm/bin/hipcc -O3 -fPIC -std=c++17 --offload-arch=gfx950 -Wno-unused-value -fno-slp-vectorize $extra -c $f -o $B/$s.o &
  while [ $(jobs -r | wc -l) -ge 4 ]; do sleep 1; done
done
wait
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 $B/*.o -o $R/scripts/diag/_bin/libias_allnoslp.so && echo built allnoslp
