#!/bin/bash
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for w in 1024 2048 4096; do
  export IAS_BN_WGS=$w
  O=$R/gpurun_out/bnw_$w; mkdir -p $O
  rocprofv3 --kernel-trace --stats -d $O -o out --output-format csv -- python3 $R/scripts/diag/time_pretrain_step.py > $O/log.txt 2>&1
  python3 - <<PY
import csv, glob
rows=[]
for f in glob.glob("$O/**/out_kernel_stats.csv", recursive=True): rows += list(csv.DictReader(open(f)))
bn=sum(float(r["TotalDurationNs"]) for r in rows if "bn_" in r["Name"])/1e6/10
tot=sum(float(r["TotalDurationNs"]) for r in rows)/1e6/10
print("bn wgs", $w, "bn ms/step", round(bn,3), "total", round(tot,2))
PY
done
