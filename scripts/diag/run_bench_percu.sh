#!/bin/bash
# headline bench with fewer resident render workgroups per CU (room for the STFT / PQMF waves on the same SIMDs)
R=$GRAFT_REPO_ROOT
cd $R
for v in 3 2 1; do
  IAS_VOICE_PERCU=$v python3 bench.py --no-cpu-baseline > gpurun_out/bench_percu_$v.json 2> gpurun_out/bench_percu_$v.err
  python3 -c "
import json; d=json.load(open('gpurun_out/bench_percu_$v.json')); print('percu', $v, d['ms_per_step'], d['ms_per_step_min'], d['roofline']['avg_launch_ms'])"
done
