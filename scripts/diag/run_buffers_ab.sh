#!/bin/bash
cd $GRAFT_REPO_ROOT
run() { name=$1; shift; python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-legs "$@" > gpurun_out/bench_$name.json 2>gpurun_out/bench_$name.err; python3 -c "
import json; d=json.load(open('gpurun_out/bench_$name.json')); print('$name', d['ms_per_step'], d['ms_per_step_min'])"; }
run b3_a --buffers 3 && run b2_a --buffers 2 && run b4_a --buffers 4 && run b5_a --buffers 5 && run b3_b --buffers 3 && run b4_b --buffers 4 && run b6 --buffers 6 && run k40 --steps 40 && run k10 --steps 10
