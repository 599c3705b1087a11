#!/bin/bash
cd $GRAFT_REPO_ROOT
run() { name=$1; shift; env "$@" python3 bench.py --workload gradstep --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/bench_$name.json 2>gpurun_out/bench_$name.err; python3 -c "
import json; d=json.load(open('gpurun_out/bench_$name.json')); print('$name', d['ms_per_step'], d['ms_per_step_min'], d['config']['phase_ms_eager'])"; }
run par_a A=1 && run ser_a IAS_BENCH_MR_SERIAL=1 && run serall_a IAS_BENCH_MR_SERIAL=1 IAS_BENCH_SERIAL_LOSSES=1 && run par_b A=1 && run ser_b IAS_BENCH_MR_SERIAL=1 && run serlosses IAS_BENCH_SERIAL_LOSSES=1
