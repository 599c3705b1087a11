#!/bin/bash
R=$GRAFT_REPO_ROOT
cd $R
for w in 1 2 3; do
  IAS_PQM_PERCU=$w python3 bench.py --no-cpu-baseline > gpurun_out/bench_pp_$w.json 2>/dev/null
  python3 -c "
import json; d=json.load(open('gpurun_out/bench_pp_$w.json')); print('pqmf wgs/cu', $w, d['ms_per_step'], d['ms_per_step_min'])"
done
for w in 2 3; do
  IAS_VOICE_PERCU=$w python3 bench.py --no-cpu-baseline > gpurun_out/bench_vp_$w.json 2>/dev/null
  python3 -c "
import json; d=json.load(open('gpurun_out/bench_vp_$w.json')); print('render wgs/cu', $w, d['ms_per_step'], d['ms_per_step_min'])"
done
