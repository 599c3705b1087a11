#!/bin/bash
# rocprofv3 kernel stats of the PQMF analysis alone (scripts/diag/time_pqmf.py): bash scripts/diag/kstats_pqmf.sh <tag> [ENV=..]
tag=${1:-x}; shift
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/kstats_pq_$tag
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for kv in "$@"; do export "$kv"; done
rocprofv3 --kernel-trace --stats -d $O/p -o out --output-format csv -- python3 $R/scripts/diag/time_pqmf.py > $O/log.txt 2>&1
python3 - <<PY
import csv, glob
for f in glob.glob("$O/p/**/out_kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        print(f"{r['Name'][:90]:90s} calls {r['Calls']:>5s} avg {float(r['AverageNs'])/1e3:8.1f} us  min {float(r['MinNs'])/1e3:8.1f}")
PY
