#!/bin/bash
# instruction-cache counters of the render alone and of the headline step's kernels (eager, un-pipelined bench so that
# the counters belong to one kernel at a time; then the pipelined default where the kernels share CUs).
# usage: bash scripts/diag/pmc_icache.sh <tag>
tag=${1:-x}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmci_$tag
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --list-avail 2>/dev/null | grep -i -E "ICACHE|SQC_|INST_CACHE|IFETCH" | head -40 > $O/avail.txt
i=0
for set in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES" "SQ_IFETCH SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_INSTS_VALU" "SQC_ICACHE_MISSES_DUPLICATE SQC_ICACHE_INPUT_VALID_READYB"; do
  i=$((i+1))
  EAGER=1 K=6 rocprofv3 --pmc $set --kernel-trace -d $O/a$i -o out --output-format csv -- python3 $R/scripts/diag/time_voice.py > $O/a$i.log 2>&1
  rocprofv3 --pmc $set --kernel-trace -d $O/b$i -o out --output-format csv -- python3 $R/bench.py --steps 10 --warmup 2 --replays 3 --no-graph --no-cpu-baseline --no-legs > $O/b$i.log 2>&1
done
python3 - > $O/summary.txt <<PY
import csv, glob, collections
for tagp, name in (("a", "render alone (time_voice.py, eager)"), ("b", "headline step, pipelined, eager launches")):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob("$O/%s*/**/*counter_collection.csv" % tagp, recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0][:50]
            if any(x in k for x in ("voice_audio", "stft2_kernel", "pqmf_analysis", "voice_env_slim", "voice_lfo_slim")):
                agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    print("==", name)
    for k in sorted(agg):
        print(k)
        for c in sorted(agg[k]):
            v = agg[k][c]
            print(f"   {c:30s} {sum(v) / len(v):18.1f}  (n={len(v)})")
PY
cat $O/avail.txt | head -20; cat $O/summary.txt
rm -rf $O/a1 $O/a2 $O/a3 $O/b1 $O/b2 $O/b3
