#!/bin/bash
cd $GRAFT_REPO_ROOT
D=inverse-audio-synthesis_amd/csrc/libias_hip_diag.so
run() { name=$1; shift; env IAS_HIP_LIB=$D "$@" python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-legs > gpurun_out/bench_$name.json 2>gpurun_out/bench_$name.err; python3 -c "
import json; d=json.load(open('gpurun_out/bench_$name.json')); k=d['roofline']['kernels']; print('$name', d['ms_per_step'], d['ms_per_step_min'], k['stft']['isolated_avg_us'])"; }
run base_a A=1 && run g384_a IAS_STFT2_WGS=384 && run g320_a IAS_STFT2_WGS=320 && run g448_a IAS_STFT2_WGS=448 && run base_b A=1 && run g384_b IAS_STFT2_WGS=384 && run g320_b IAS_STFT2_WGS=320 && run g448_b IAS_STFT2_WGS=448 && run base_c A=1 && run g384_c IAS_STFT2_WGS=384
