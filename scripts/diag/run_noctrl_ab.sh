#!/bin/bash
# same-box A/B/A/B: the headline step with and without the control pass in the loop
cd $GRAFT_REPO_ROOT
run() { name=$1; shift; env "$@" python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-legs > gpurun_out/bench_$name.json 2>gpurun_out/bench_$name.err; python3 -c "
import json; d=json.load(open('gpurun_out/bench_$name.json')); print('$name', d['ms_per_step'], d['ms_per_step_min'])"; }
run ctrl_a A=1 && run noctrl_a IAS_BENCH_NOCTRL=1 && run ctrl_b A=1 && run noctrl_b IAS_BENCH_NOCTRL=1 && python3 scripts/diag/time_overlap.py
