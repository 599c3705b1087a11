#!/bin/bash
# GRBM_GUI_ACTIVE (GPU-busy cycles) per kernel against its traced duration -> effective shader clock per kernel
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
mkdir -p $R/gpurun_out/pmc_clock
rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace -d $R/gpurun_out/pmc_clock -o out --output-format csv -- \
  python3 $R/bench.py --steps 3 --warmup 1 --no-graph --no-pipeline --no-cpu-baseline --no-legs > $R/gpurun_out/pmc_clock/log.txt 2>&1
python3 - <<PY
import csv, glob, collections
dur = {}
for f in glob.glob("$R/gpurun_out/pmc_clock/**/out_kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        dur[r["Dispatch_Id"]] = (r["Kernel_Name"].split("(")[0][:44], int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
agg = collections.defaultdict(list)
for f in glob.glob("$R/gpurun_out/pmc_clock/**/out_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == "GRBM_GUI_ACTIVE" and r["Dispatch_Id"] in dur:
            n, d = dur[r["Dispatch_Id"]]
            agg[n].append((float(r["Counter_Value"]), d))
for n, v in agg.items():
    c = sum(a for a, _ in v) / len(v); d = sum(b for _, b in v) / len(v)
    print(f"{n:44s} cycles {c:12.0f}  dur {d/1e3:8.1f} us  -> {c/d:6.2f} GHz (x1, summed over XCDs: /8 = {c/d/8:.2f})")
PY
