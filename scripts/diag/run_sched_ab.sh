#!/bin/bash
# issue-order knobs of the headline schedule, re-measured with the PQMF and control kernels beside the render; same box
cd $GRAFT_REPO_ROOT
run() { name=$1; shift; env "$@" > gpurun_out/bench_$name.json 2>gpurun_out/bench_$name.err; python3 -c "
import json; d=json.load(open('gpurun_out/bench_$name.json')); print('$name', d['ms_per_step'], d['ms_per_step_min'])"; }
B="python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-legs"
run base_a A=1 $B && run nodefer A=1 $B --no-defer-consumers && run pqmf_first IAS_BENCH_CONSUMERS=pqmf_first $B && run stft_first IAS_BENCH_CONSUMERS=stft_first $B && run base_b A=1 $B && run prio_render IAS_BENCH_PRIO=render $B && run prio_cons IAS_BENCH_PRIO=consumers $B && run red_ctrl IAS_BENCH_REDUCE_STREAM=control $B && run base_c A=1 $B
