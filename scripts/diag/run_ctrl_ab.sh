#!/bin/bash
# control pass forms alone, then the headline step with each (diag library selects libm / fused), same box
cd $GRAFT_REPO_ROOT
D=inverse-audio-synthesis_amd/csrc/libias_hip_diag.so
python3 scripts/diag/time_ctrl.py && IAS_HIP_LIB=$D IAS_VOICE_CTRL=libm python3 scripts/diag/time_ctrl.py && IAS_HIP_LIB=$D IAS_VOICE_CTRL=fused python3 scripts/diag/time_ctrl.py || exit 1
run() { name=$1; shift; env "$@" python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-legs > gpurun_out/bench_$name.json 2>gpurun_out/bench_$name.err; python3 -c "
import json; d=json.load(open('gpurun_out/bench_$name.json')); print('$name', d['ms_per_step'], d['ms_per_step_min'])"; }
run slim_a A=1 && run libm_a IAS_HIP_LIB=$D IAS_VOICE_CTRL=libm && run fused_a IAS_HIP_LIB=$D IAS_VOICE_CTRL=fused && run noctrl_a IAS_BENCH_NOCTRL=1 && run slim_b A=1 && run libm_b IAS_HIP_LIB=$D IAS_VOICE_CTRL=libm && run noctrl_b IAS_BENCH_NOCTRL=1
