#!/bin/bash
# PQMF analysis A/B on the GPU box: MFMA paths vs the VALU kernels (IAS_PQMF_VALU=1), N=3 headline and N=64 configs[4]
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pqmf_ab.txt
: > $O
run() { env "$@" python3 $R/scripts/diag/time_pqmf.py 2>/dev/null | grep pqmf | sed "s/\$/  [$*]/" >> $O; }
run N=3
run N=3 IAS_PQM_PERCU=1
run N=3 IAS_PQM_PERCU=2
run N=3 IAS_PQM_TILED=1
run N=3 IAS_PQMF_VALU=1
run N=64 B=64
run N=64 B=64 IAS_PQMF_VALU=1
cat $O
