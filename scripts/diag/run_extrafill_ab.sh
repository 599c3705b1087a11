#!/bin/bash
# is the render queue the step's critical chain?  n extra 90 KB memset nodes in front of every render; same box
cd $GRAFT_REPO_ROOT
run() { name=$1; shift; env "$@" python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-legs > gpurun_out/bench_$name.json 2>gpurun_out/bench_$name.err; python3 -c "
import json; d=json.load(open('gpurun_out/bench_$name.json')); print('$name', d['ms_per_step'], d['ms_per_step_min'])"; }
run fill0_a IAS_BENCH_EXTRA_FILL=0 && run fill1_a IAS_BENCH_EXTRA_FILL=1 && run fill2_a IAS_BENCH_EXTRA_FILL=2 && run fill0_b IAS_BENCH_EXTRA_FILL=0 && run fill1_b IAS_BENCH_EXTRA_FILL=1 && run fill4 IAS_BENCH_EXTRA_FILL=4
