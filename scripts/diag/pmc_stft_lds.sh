#!/bin/bash
# LDS counters of stft_kernel per variant (PARTS=loss|mel|raw of scripts/diag/time_stft_parts.py).  usage: bash scripts/diag/pmc_stft_lds.sh <tag>
tag=${1:-x}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmcl_$tag
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for part in loss raw; do
  export PARTS=$part
  rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_BUSY_CYCLES --kernel-trace -d $O/$part -o out --output-format csv -- python3 $R/scripts/diag/time_stft_parts.py > $O/$part.log 2>&1
done
python3 - > $O/summary.txt <<PY
import csv, glob, collections
for part in ("loss", "raw"):
    agg = collections.defaultdict(list)
    for f in glob.glob("$O/" + part + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "stft_kernel" in r["Kernel_Name"]:
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    print(part)
    for c in sorted(agg):
        print(f"   {c:28s} {sum(agg[c]) / len(agg[c]):18.1f}  (n={len(agg[c])})")
PY
