#!/bin/bash
# PMC passes over one bench step schedule (no graph, no pipeline): per-kernel SQ counters -> gpurun_out/pmc_<tag>/
# usage (on the GPU box): bash scripts/diag/pmc_stage.sh <tag>
set -e
tag=${1:-x}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU" \
           "SQ_INSTS_SMEM SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_LDS" \
           "SQ_INST_CYCLES_SMEM SQ_ACTIVE_INST_SCA SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY"; do
  name=$(echo $set | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $set --kernel-trace -d $R/gpurun_out/pmc_$tag/$name -o out --output-format csv -- \
    python3 $R/bench.py --steps 3 --warmup 1 --no-graph --no-pipeline --no-cpu-baseline --no-legs > $R/gpurun_out/pmc_$tag/$name.log 2>&1
done
python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(lambda: collections.defaultdict(int))
for f in glob.glob("$R/gpurun_out/pmc_$tag/*/*counter_collection.csv") + glob.glob("$R/gpurun_out/pmc_$tag/*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][:48]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[k][r["Counter_Name"]] += 1
for k in sorted(agg):
    print(k)
    for c in sorted(agg[k]):
        print(f"   {c:26s} {agg[k][c] / cnt[k][c]:16.1f}  (n={cnt[k][c]})")
PY
