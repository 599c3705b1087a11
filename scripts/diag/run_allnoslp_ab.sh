#!/bin/bash
# the in-tree library against whole-library build variants (scripts/diag/build_allnoslp.sh, build_allflags.sh) on all workloads,
# interleaved on one box:  bash scripts/diag/run_allnoslp_ab.sh [variant ...]   (default: product allnoslp product allnoslp)
R=$GRAFT_REPO_ROOT
[ $# -eq 0 ] && set -- product allnoslp product allnoslp
for v in "$@"; do
  if [ $v = product ]; then unset IAS_HIP_LIB; else export IAS_HIP_LIB=$R/scripts/diag/_bin/libias_$v.so; fi
  p=$(GRAPH=1 STEPS=10 python3 $R/scripts/diag/time_pretrain_step.py 2>&1 | tail -1 | cut -c49-62)
  h=$(python3 $R/bench.py --no-legs --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; j=json.loads(sys.stdin.read()); k=j['roofline']['kernels']; print(j['ms_per_step'], k['render']['isolated_avg_us'], k['pqmf']['isolated_avg_us'], k['stft']['isolated_avg_us'])")
  g=$(python3 $R/bench.py --workload gradstep --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; j=json.loads(sys.stdin.read()); print(j['ms_per_step'])")
  v1=$(python3 $R/bench.py --workload vicreg --batch 1024 --steps 10 --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; j=json.loads(sys.stdin.read()); print(j['ms_per_step'])")
  echo "$v: pretrain $p | headline (render, pqmf, stft us) $h | gradstep $g | vicreg1024 $v1"
done
