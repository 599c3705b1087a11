#!/bin/bash
# SQ / LDS / TCP counters of the audio-rate render kernel alone (scripts/diag/time_voice.py, eager launches).
# usage (GPU box): bash scripts/diag/pmc_voice.sh <tag>       -> gpurun_out/pmcv_<tag>/summary.txt
tag=${1:-x}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmcv_$tag
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU" \
           "SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM" \
           "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE" "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum" "TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  EAGER=1 K=6 rocprofv3 --pmc $set --kernel-trace -d $O/p$i -o out --output-format csv -- python3 $R/scripts/diag/time_voice.py > $O/p$i.log 2>&1
done
EAGER=1 K=20 rocprofv3 --kernel-trace --stats -d $O/stats -o out --output-format csv -- python3 $R/scripts/diag/time_voice.py > $O/stats.log 2>&1
python3 - > $O/summary.txt <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$O/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][:60]
        if "voice_audio" in k:
            agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(agg):
    print(k)
    for c in sorted(agg[k]):
        v = agg[k][c]
        print(f"   {c:26s} {sum(v) / len(v):18.1f}  (n={len(v)})")
for f in glob.glob("$O/stats/**/out_kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        print(f"{r['Name'][:70]:70s} calls {r['Calls']:>5s} avg {float(r['AverageNs'])/1e3:8.1f} us  min {float(r['MinNs'])/1e3:8.1f}")
PY
cat $O/summary.txt
