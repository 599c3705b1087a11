#!/bin/bash
# Same-box A/B of the round-4 tree (a git worktree of a63291d built under scripts/diag/_bin/r04tree) against the current
# tree: the driver's command (headline step, legs).  bash scripts/diag/run_r04_vs_r05.sh
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
out=$R/gpurun_out/r04_vs_r05.txt; : > $out
for rep in 1 2; do
  for tree in r04 r05; do
    if [ $tree = r04 ]; then cd $R/scripts/diag/_bin/r04tree; else cd $R; fi
    python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $R/gpurun_out/ab_$tree.json 2>/dev/null
    python3 - $tree $R/gpurun_out/ab_$tree.json >> $out <<PY
import json, sys
j = json.load(open(sys.argv[2]))
k = j["roofline"]["kernels"]
l = j["legs"]
print(sys.argv[1], "step", j["ms_per_step"], "min", j["ms_per_step_min"], "iso", {n: k[n]["isolated_avg_us"] for n in k},
      "v128", l["vicreg128"]["ms_per_step"], "v1024", l["vicreg1024"]["ms_per_step"], "gradstep", l["gradstep"]["ms_per_step"],
      "pretrain", l["pretrain"]["ms_per_step"])
PY
  done
done
cat $out
