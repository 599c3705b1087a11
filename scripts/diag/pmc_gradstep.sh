#!/bin/bash
# SQ counters per kernel of the configs[4] gradient step (bench.py --workload gradstep, eager launches), each counter set
# in a run of its own with --kernel-trace only (MI355X_MICROARCH.md: never --pmc together with the runtime traces), plus
# the rocprofv3 --stats durations of the same command.  -> gpurun_out/pmcgs_<tag>/summary.txt, counters_gradstep.json
# usage: bash scripts/diag/pmc_gradstep.sh <tag>
tag=${1:-x}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmcgs_$tag
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" \
           "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" \
           "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace -d $O/p$i -o out --output-format csv -- python3 $R/bench.py --workload gradstep --steps 3 --warmup 1 --replays 2 --no-cpu-baseline --no-graph > $O/p$i.log 2>&1
done
rocprofv3 --kernel-trace --stats -d $O/stats -o out --output-format csv -- python3 $R/bench.py --workload gradstep --steps 5 --warmup 2 --replays 3 --no-cpu-baseline --no-graph > $O/stats.log 2>&1
python3 - <<PY
import csv, glob, collections, json, os, sys
sys.path.insert(0, "$R/scripts")
from make_counters import kernel_sources_sha16
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$O/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        if k.startswith("at::") or k.startswith("__amd"):
            continue
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
dur = {}
for f in glob.glob("$O/stats/**/out_kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Name"].split("(")[0].replace("void ", "")
        dur[k] = {"calls": int(r["Calls"]), "avg_us": float(r["AverageNs"]) / 1e3, "min_us": float(r["MinNs"]) / 1e3}
out = {"_method": "rocprofv3 --pmc (three counter sets, each in a run of its own, --kernel-trace only) over `bench.py --workload gradstep "
                  "--no-graph` (B = 64 x 176400, eager launches), averages per launch; SQ_* are sums over all CUs; avg_us / min_us from "
                  "rocprofv3 --kernel-trace --stats of the same command (in-step durations: the kernels share the GPU)",
       "_round": "$tag", "_source_sha16": kernel_sources_sha16("$R")}
with open("$O/summary.txt", "w") as fh:
    for k in sorted(agg):
        out[k] = {c: sum(v) / len(v) for c, v in agg[k].items()}
        out[k].update(dur.get(k, {}))
        fh.write(k + "\n")
        for c in sorted(out[k]):
            fh.write(f"   {c:28s} {out[k][c]:18.1f}\n")
json.dump(out, open("$O/counters_gradstep.json", "w"), indent=1)
PY
rm -rf $O/p1 $O/p2 $O/p3 $O/stats
cat $O/summary.txt | head -60
