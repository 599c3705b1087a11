#!/bin/bash
# Same-box A/B of the round-3 tree (a git worktree of ee5cfd5 built under scripts/diag/_bin/r03tree) against the current
# tree: headline step, VICReg legs, pretraining step.  bash scripts/diag/run_r03_vs_r04.sh
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
out=$R/gpurun_out/r03_vs_r04.txt; : > $out
for rep in 1 2; do
  for tree in r03 r04; do
    if [ $tree = r03 ]; then cd $R/scripts/diag/_bin/r03tree; else cd $R; fi
    python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $R/gpurun_out/ab_$tree.json 2>/dev/null
    python3 - $tree $R/gpurun_out/ab_$tree.json >> $out <<PY
import json, sys
j = json.load(open(sys.argv[2]))
k = j["roofline"]["kernels"]
l = j["legs"]
print(sys.argv[1], "step", j["ms_per_step"], "min", j["ms_per_step_min"], "iso", {n: k[n]["isolated_avg_us"] for n in k},
      "gram128", l["vicreg128"]["roofline"]["avg_launch_ms"], l["vicreg128"]["roofline"]["frac"], "v128", l["vicreg128"]["ms_per_step"],
      "gram1024", l["vicreg1024"]["roofline"]["frac"], "gradstep", l["gradstep"]["ms_per_step"], "pretrain", l["pretrain"]["ms_per_step"])
PY
  done
done
cat $out
