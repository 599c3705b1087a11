#!/bin/bash
# headline bench with fewer resident render workgroups per CU (room for the STFT / PQMF waves on the same SIMDs)
R=$GRAFT_REPO_ROOT
cd $R
for v in 3 2; do
  for b in 3; do
  IAS_VOICE_PERCU=$v python3 bench.py --no-cpu-baseline --buffers $b > gpurun_out/bench_percu_$v$b.json 2> gpurun_out/bench_percu_$v$b.err
  python3 -c "
import json; d=json.load(open('gpurun_out/bench_percu_$v$b.json')); print('percu', $v, 'buffers', $b, d['ms_per_step'], d['ms_per_step_min'], d['roofline']['avg_launch_ms'])"
  done
done
