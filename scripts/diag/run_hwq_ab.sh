#!/bin/bash
# ROCm's GPU_MAX_HW_QUEUES (default 4): more hardware queues for the captured steps' parallel branches?  same box, A/B/A/B
cd $GRAFT_REPO_ROOT
run() { name=$1; shift; env "$@" python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/bench_$name.json 2>gpurun_out/bench_$name.err; python3 -c "
import json; d=json.load(open('gpurun_out/bench_$name.json')); l=d['legs']; print('$name', 'step', d['ms_per_step'], d['ms_per_step_min'], 'gradstep', l['gradstep']['ms_per_step'], 'pretrain', l['pretrain']['ms_per_step'], 'v1024', l['vicreg1024']['ms_per_step'])"; }
run q4_a GPU_MAX_HW_QUEUES=4 && run q8_a GPU_MAX_HW_QUEUES=8 && run q4_b GPU_MAX_HW_QUEUES=4 && run q8_b GPU_MAX_HW_QUEUES=8 && run q6 GPU_MAX_HW_QUEUES=6 && run q2 GPU_MAX_HW_QUEUES=2
