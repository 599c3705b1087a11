#!/bin/bash
R=$GRAFT_REPO_ROOT
cd $R
run() { name=$1; shift; env "$@" python3 bench.py --no-cpu-baseline > gpurun_out/bench_$name.json 2>gpurun_out/bench_$name.err; python3 -c "
import json; d=json.load(open('gpurun_out/bench_$name.json')); print('$name', d['ms_per_step'], d['ms_per_step_min'], d['config']['loss'])"; }
run reduce_aside A=1
run reduce_inline IAS_BENCH_REDUCE_INLINE=1
run reduce_aside2 A=1
