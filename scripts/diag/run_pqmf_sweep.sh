#!/bin/bash
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pqmf_sweep.txt
: > $O
for v in ${VARIANTS:-noload_nostore}; do
  for pc in ${PCS:-1 2 3 5}; do
    env N=3 IAS_PQM_PERCU=$pc IAS_HIP_LIB=$R/scripts/diag/_bin/libias_pq_$v.so python3 $R/scripts/diag/time_pqmf.py 2>/dev/null | grep pqmf | sed "s/\$/  [$v percu=$pc]/" >> $O
  done
done
cat $O
