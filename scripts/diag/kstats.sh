#!/bin/bash
# rocprofv3 kernel stats of time_stages.py (true kernel durations, free of host launch gaps)
# usage on the GPU box: bash scripts/diag/kstats.sh <tag>
tag=${1:-x}
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/kstats_$tag -o out --output-format csv -- python3 $R/scripts/diag/time_stages.py > $R/gpurun_out/kstats_$tag.log 2>&1
python3 - <<PY
import csv, glob
for f in glob.glob("$R/gpurun_out/kstats_$tag/**/out_kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        print(f"{r['Name'][:60]:60s} calls {r['Calls']:>5s} avg {float(r['AverageNs'])/1e3:8.1f} us  min {float(r['MinNs'])/1e3:8.1f}")
PY
