#!/bin/bash
R=$GRAFT_REPO_ROOT
cd $R
for w in ${WGS:-256 384 512 768 1024}; do
  IAS_STFT_WGS=$w python3 bench.py --no-cpu-baseline > gpurun_out/bench_sw_$w.json 2>/dev/null
  python3 -c "
import json; d=json.load(open('gpurun_out/bench_sw_$w.json')); print('stft wgs', $w, d['ms_per_step'], d['ms_per_step_min'])"
done
