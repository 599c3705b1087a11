#!/bin/bash
# same-box A/B/A/B of the configs[4] gradient step: baseline tree (scripts/diag/_bin/basetree, see run_tree_ab.sh) against
# this tree.   bash scripts/diag/run_gradstep_tree_ab.sh
R=$GRAFT_REPO_ROOT
for rep in 1 2; do
  for tree in base new; do
    if [ $tree = base ]; then cd $R/scripts/diag/_bin/basetree; else cd $R; fi
    python3 bench.py --workload gradstep --steps 20 --warmup 3 --no-cpu-baseline > $R/gpurun_out/gs_$tree.json 2> $R/gpurun_out/gs_$tree.err
    python3 -c "
import json; d=json.load(open('$R/gpurun_out/gs_$tree.json')); print('$tree', d['ms_per_step'], d.get('ms_per_step_min'))"
  done
done
