#!/bin/bash
# the mel-STFT kernel's waves per workgroup / persistent grid, re-measured now that the PQMF runs inside the render
# (diagnostic library: the product library has no switches); same box
cd $GRAFT_REPO_ROOT
D=inverse-audio-synthesis_amd/csrc/libias_hip_diag.so
run() { name=$1; shift; env IAS_HIP_LIB=$D "$@" python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-legs > gpurun_out/bench_$name.json 2>gpurun_out/bench_$name.err; python3 -c "
import json; d=json.load(open('gpurun_out/bench_$name.json')); k=d['roofline']['kernels']; print('$name', d['ms_per_step'], d['ms_per_step_min'], k['stft']['isolated_avg_us'])"; }
run base_a A=1 && run w4 IAS_STFT2_WAVES=4 && run w5 IAS_STFT2_WAVES=5 && run w10 IAS_STFT2_WAVES=10 && run base_b A=1 && run g256 IAS_STFT2_WGS=256 && run g384 IAS_STFT2_WGS=384 && run g768 IAS_STFT2_WGS=768 && run base_c A=1
