#!/bin/bash
# headline bench with fewer resident render workgroups per CU (room for the STFT / PQMF waves on the same SIMDs); the switch
# lives in the diagnostic library.   bash scripts/diag/run_bench_percu.sh
R=$GRAFT_REPO_ROOT
cd $R
D=$R/inverse-audio-synthesis_amd/csrc/libias_hip_diag.so
for rep in 1 2; do
for v in 3 2; do
  IAS_HIP_LIB=$D IAS_VOICE_PERCU=$v python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-legs > gpurun_out/bench_percu_$v.json 2> gpurun_out/bench_percu_$v.err
  python3 -c "
import json; d=json.load(open('gpurun_out/bench_percu_$v.json')); k=d['roofline']['kernels']; print('percu', $v, d['ms_per_step'], d['ms_per_step_min'], {n: k[n]['isolated_avg_us'] for n in k})"
done
done
