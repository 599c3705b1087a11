#!/bin/bash
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pqmf_variants.txt
: > $O
for v in ${VARIANTS:-nomfma nostore nomfma_nostore}; do
  for n in "N=3" "N=64 B=64"; do
    env $n IAS_HIP_LIB=$R/scripts/diag/_bin/libias_pq_$v.so python3 $R/scripts/diag/time_pqmf.py 2>/dev/null | grep pqmf | sed "s/\$/  [$v]/" >> $O
  done
done
cat $O
