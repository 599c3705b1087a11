#!/bin/bash
# the modulated PQMF(3) kernel in its <= 56-VGPR stream form (product) against the 96-VGPR window form (diagnostic library,
# IAS_PQMF_MOD_WINDOW=1): alone, and inside the headline step; same box, alternating
cd $GRAFT_REPO_ROOT
D=inverse-audio-synthesis_amd/csrc/libias_hip_diag.so
run() { name=$1; shift; env "$@" python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-legs > gpurun_out/bench_$name.json 2>gpurun_out/bench_$name.err; python3 -c "
import json; d=json.load(open('gpurun_out/bench_$name.json')); k=d['roofline']['kernels']; print('$name', d['ms_per_step'], d['ms_per_step_min'], {n: (k[n]['isolated_avg_us'], k[n]['in_step_avg_us']) for n in k})"; }
run stream_a A=1 && run window_a IAS_HIP_LIB=$D IAS_PQMF_MOD_WINDOW=1 && run stream_b A=1 && run window_b IAS_HIP_LIB=$D IAS_PQMF_MOD_WINDOW=1 && run stream_c A=1
