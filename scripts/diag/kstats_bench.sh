#!/bin/bash
# rocprofv3 kernel stats of the default (pipelined, graph-replayed) bench step: per-kernel in-step durations.
# usage on the GPU box: bash scripts/diag/kstats_bench.sh <tag> [ENV=.. ...] [-- bench args]
tag=${1:-x}; shift
R=$GRAFT_REPO_ROOT
args="--no-cpu-baseline"
while [ $# -gt 0 ]; do case "$1" in --) shift; args="$args $*"; break;; *) export "$1"; shift;; esac; done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/kb_$tag -o out --output-format csv -- python3 $R/bench.py $args > $R/gpurun_out/kb_$tag.log 2>&1
python3 - <<PY
import csv, glob
for f in glob.glob("$R/gpurun_out/kb_$tag/**/out_kernel_stats.csv", recursive=True):
    rows = sorted(csv.DictReader(open(f)), key=lambda r: -float(r["TotalDurationNs"]))
    for r in rows[:9]:
        print(f"{r['Name'][:58]:58s} calls {r['Calls']:>6s} avg {float(r['AverageNs'])/1e3:7.1f} us min {float(r['MinNs'])/1e3:7.1f} tot% {r['Percentage']}")
PY
tail -1 $R/gpurun_out/kb_$tag.log | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('ms_per_step', d['ms_per_step'], d['ms_per_step_min'])"
