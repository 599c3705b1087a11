#!/bin/bash
R=$GRAFT_REPO_ROOT
cd $R
for v in parallel pqmf_first stft_first; do
  for nb in 2 3; do
  IAS_BENCH_CONSUMERS=$v python3 bench.py --no-cpu-baseline --buffers $nb > gpurun_out/bench_order_$v$nb.json 2> gpurun_out/bench_order_$v$nb.err
  python3 -c "
import json; d=json.load(open('gpurun_out/bench_order_$v$nb.json')); print('$v buffers $nb', d['ms_per_step'], d['ms_per_step_min'])"
  done
done
