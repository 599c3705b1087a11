#!/bin/bash
# libias variants with parts of the MFMA PQMF kernels compiled out (diagnostics): scripts/diag/_bin/libias_pq_<v>.so
# usage: bash scripts/diag/build_pqmf_variants.sh "name:-DFLAG%-DFLAG2" ...
cd /root/repo/inverse-audio-synthesis_amd/csrc
mkdir -p ../../scripts/diag/_bin
OBJS=$(ls *.o | grep -v pqmf_kernels.o)
for v in "$@"; do
  name=${v%%:*}; flags=${v#*:}; flags=${flags//%/ }
  /opt/rocm/bin/hipcc -O3 -fPIC -std=c++17 --offload-arch=gfx950 -Wno-unused-value -fno-slp-vectorize $flags -c pqmf_kernels.hip -o /tmp/pqmf_$name.o &&
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 $OBJS /tmp/pqmf_$name.o -o ../../scripts/diag/_bin/libias_pq_$name.so
done
ls ../../scripts/diag/_bin/ | grep pq_
