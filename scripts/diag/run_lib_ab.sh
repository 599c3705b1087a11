#!/bin/bash
# same-box A/B/A/B of two builds of the product library on the headline step: scripts/diag/_bin/libias_hip_base.so (built
# from the tree one wants to compare against) and the in-tree library.   bash scripts/diag/run_lib_ab.sh
cd $GRAFT_REPO_ROOT
run() { name=$1; shift; env "$@" python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-legs > gpurun_out/bench_$name.json 2>gpurun_out/bench_$name.err; python3 -c "
import json; d=json.load(open('gpurun_out/bench_$name.json')); k=d['roofline']['kernels']; print('$name', d['ms_per_step'], d['ms_per_step_min'], {n: k[n]['isolated_avg_us'] for n in k})"; }
B=$GRAFT_REPO_ROOT/scripts/diag/_bin/libias_hip_base.so
run base_a IAS_HIP_LIB=$B && run new_a A=1 && run base_b IAS_HIP_LIB=$B && run new_b A=1
