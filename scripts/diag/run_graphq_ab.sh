#!/bin/bash
# ROCm's hipGraph executor spreads a graph's parallel branches over DEBUG_HIP_FORCE_GRAPH_QUEUES streams (default 4) that
# map onto GPU_MAX_HW_QUEUES hardware queues (default 4): the gradient step's graph has 5-6 parallel branches, the headline
# step's 5.  Same box, headline + gradient step per setting.   bash scripts/diag/run_graphq_ab.sh
cd $GRAFT_REPO_ROOT
run() { name=$1; shift; env "$@" python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-legs > gpurun_out/bench_gq_$name.json 2>gpurun_out/bench_gq_$name.err
  env "$@" python3 bench.py --workload gradstep --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/bench_gqg_$name.json 2>gpurun_out/bench_gqg_$name.err
  python3 -c "
import json; d=json.load(open('gpurun_out/bench_gq_$name.json')); g=json.load(open('gpurun_out/bench_gqg_$name.json')); print('$name', 'step', d['ms_per_step'], d['ms_per_step_min'], 'gradstep', g['ms_per_step'], g.get('ms_per_step_min'))"; }
run default A=1 && run g6 DEBUG_HIP_FORCE_GRAPH_QUEUES=6 && run g8 DEBUG_HIP_FORCE_GRAPH_QUEUES=8 && run g8h8 DEBUG_HIP_FORCE_GRAPH_QUEUES=8 GPU_MAX_HW_QUEUES=8 && run g6h6 DEBUG_HIP_FORCE_GRAPH_QUEUES=6 GPU_MAX_HW_QUEUES=6 && run g2 DEBUG_HIP_FORCE_GRAPH_QUEUES=2 && run default_b A=1
