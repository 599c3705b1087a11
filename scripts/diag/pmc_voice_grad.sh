#!/bin/bash
# SQ / LDS counters of the audio-rate Voice backward kernels (scripts/diag/time_voice_grad.py).
# usage (GPU box): bash scripts/diag/pmc_voice_grad.sh <tag>       -> gpurun_out/pmcvg_<tag>/summary.txt
tag=${1:-x}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmcvg_$tag
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" \
           "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VALU" \
           "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace -d $O/p$i -o out --output-format csv -- python3 $R/scripts/diag/time_voice_grad.py > $O/p$i.log 2>&1
done
python3 - > $O/summary.txt <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$O/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][:60]
        if "voice_grad" in k:
            agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(agg):
    print(k)
    for c in sorted(agg[k]):
        v = agg[k][c]
        print(f"   {c:26s} {sum(v) / len(v):18.1f}  (n={len(v)})")
PY
rm -rf $O/p*
cat $O/summary.txt
