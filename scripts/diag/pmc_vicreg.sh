#!/bin/bash
# SQ counters of the VICReg Gram kernel (bench.py --workload vicreg, stage-1 launches).  usage: bash scripts/diag/pmc_vicreg.sh <tag> [batch]
tag=${1:-x}; batch=${2:-128}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmcg_$tag
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES" \
           "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT" \
           "SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU" \
           "GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace -d $O/p$i -o out --output-format csv -- python3 $R/bench.py --workload vicreg --batch $batch --steps 4 --warmup 1 --no-cpu-baseline --no-graph > $O/p$i.log 2>&1
done
rocprofv3 --kernel-trace --stats -d $O/stats -o out --output-format csv -- python3 $R/bench.py --workload vicreg --batch $batch --steps 10 --warmup 2 --no-cpu-baseline --no-graph > $O/stats.log 2>&1
python3 - > $O/summary.txt <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$O/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][:60]
        if "vicreg" in k:
            agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(agg):
    print(k)
    for c in sorted(agg[k]):
        v = agg[k][c]
        print(f"   {c:28s} {sum(v) / len(v):18.1f}  (n={len(v)})")
for f in glob.glob("$O/stats/**/out_kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "vicreg" in r["Name"]:
            print(f"{r['Name'][:70]:70s} calls {r['Calls']:>5s} avg {float(r['AverageNs'])/1e3:8.1f} us  min {float(r['MinNs'])/1e3:8.1f}")
PY
cat $O/summary.txt
