#!/bin/bash
# A/B of csrc/spectral_kernels.hip build variants (scripts/diag/build_variants.sh) on the headline step and the gradient step:
#   bash scripts/diag/run_spectral_ab.sh <variant> ...   ("product" = the in-tree library)
R=$GRAFT_REPO_ROOT
for round in 1 2; do
  for v in "$@"; do
    if [ $v = product ]; then unset IAS_HIP_LIB; else export IAS_HIP_LIB=$R/scripts/diag/_bin/libias_$v.so; fi
    h=$(python3 $R/bench.py --no-legs --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; j=json.loads(sys.stdin.read()); print(j['ms_per_step'], j['roofline']['kernels']['stft']['isolated_avg_us'])")
    g=$(python3 $R/bench.py --workload gradstep --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; j=json.loads(sys.stdin.read()); print(j['ms_per_step'])")
    echo "$v: headline ms/step, stft us = $h ; gradstep ms = $g"
  done
done
