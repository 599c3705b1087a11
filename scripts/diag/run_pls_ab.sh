#!/bin/bash
cd $GRAFT_REPO_ROOT
run() { name=$1; shift; env "$@" python3 bench.py --workload gradstep --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/bench_$name.json 2>gpurun_out/bench_$name.err; python3 -c "
import json; d=json.load(open('gpurun_out/bench_$name.json')); print('$name', d['ms_per_step'], d['ms_per_step_min'])"; }
run share0_a IAS_BENCH_PLS_SHARE=0 && run share1_a IAS_BENCH_PLS_SHARE=1 && run share2_a IAS_BENCH_PLS_SHARE=2 && run own_a IAS_BENCH_PLS_SHARE=-1 && run share0_b IAS_BENCH_PLS_SHARE=0 && run share1_b IAS_BENCH_PLS_SHARE=1 && run share2_b IAS_BENCH_PLS_SHARE=2 && run own_b IAS_BENCH_PLS_SHARE=-1
