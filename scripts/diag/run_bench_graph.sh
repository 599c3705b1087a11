#!/bin/bash
# headline bench: graph vs eager, consumer issue order, pipeline depth (same box)
R=$GRAFT_REPO_ROOT
cd $R
run() { name=$1; shift; python3 bench.py --no-cpu-baseline "$@" > gpurun_out/bench_$name.json 2>gpurun_out/bench_$name.err; python3 -c "
import json; d=json.load(open('gpurun_out/bench_$name.json')); print('$name', d['ms_per_step'], d['ms_per_step_min'], d['config'].get('launch'))"; }
run graph_defer_b2
run graph_defer_b3 --buffers 3
run graph_nodefer_b2 --no-defer-consumers
run graph_nodefer_b3 --no-defer-consumers --buffers 3
run eager_defer_b3 --no-graph --buffers 3
run eager_nodefer_b3 --no-graph --buffers 3 --no-defer-consumers
