#!/bin/bash
# rocprofv3 kernel stats of any python script: bash scripts/diag/kstats_script.sh <tag> <script.py> [ENV=.. ...]
tag=$1; script=$2; shift 2
R=$GRAFT_REPO_ROOT
for kv in "$@"; do export "$kv"; done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/ks_$tag -o out --output-format csv -- python3 $R/$script > $R/gpurun_out/ks_$tag.log 2>&1
tail -2 $R/gpurun_out/ks_$tag.log
python3 - <<PY
import csv, glob
for f in glob.glob("$R/gpurun_out/ks_$tag/**/out_kernel_stats.csv", recursive=True):
    rows = sorted(csv.DictReader(open(f)), key=lambda r: -float(r["TotalDurationNs"]))
    for r in rows[:16]:
        print(f"{r['Name'][:70]:70s} calls {r['Calls']:>6s} avg {float(r['AverageNs'])/1e3:7.1f} us min {float(r['MinNs'])/1e3:7.1f} tot% {r['Percentage']}")
PY
