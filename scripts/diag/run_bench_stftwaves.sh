#!/bin/bash
R=$GRAFT_REPO_ROOT
cd $R
for w in 4 8; do
  IAS_STFT_WAVES=$w python3 bench.py --no-cpu-baseline > gpurun_out/bench_swv_$w.json 2>/dev/null
  python3 -c "
import json; d=json.load(open('gpurun_out/bench_swv_$w.json')); print('stft waves/wg', $w, d['ms_per_step'], d['ms_per_step_min'])"
  IAS_STFT_WAVES=$w python3 scripts/diag/time_stft_parts.py 2>&1 | grep loss
done
