#!/bin/bash
# SQ counters of the STFT kernel alone (scripts/diag/time_stft_parts.py; PARTS=loss|mel|raw selects one variant).
# usage: bash scripts/diag/pmc_stft.sh <tag> [ENV=.. ...]
tag=${1:-x}; shift
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmcs_$tag
mkdir -p $O
for kv in "$@"; do export "$kv"; done
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES" \
           "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT" \
           "SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU" \
           "GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace -d $O/p$i -o out --output-format csv -- python3 $R/scripts/diag/time_stft_parts.py > $O/p$i.log 2>&1
done
python3 - > $O/summary.txt <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$O/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][:60]
        if "stft_kernel" in k or "stft_mfma_kernel" in k or "stft2_kernel" in k:
            agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(agg):
    print(k)
    for c in sorted(agg[k]):
        v = agg[k][c]
        print(f"   {c:28s} {sum(v) / len(v):18.1f}  (n={len(v)})")
PY
cat $O/summary.txt
