#!/bin/bash
# rocprofv3 kernel stats of the configs[4] gradient step (bench.py --workload gradstep, eager launches): per-step table
tag=${1:-x}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/kstats_gs_$tag
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/prof -o out --output-format csv -- python3 $R/bench.py --workload gradstep --steps 10 --warmup 2 --no-cpu-baseline --no-graph > $O/prof.log 2>&1
python3 - <<PY
import csv, glob
rows = []
for f in glob.glob("$O/prof/**/out_kernel_stats.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
# launches of the render kernel = number of steps executed (warm-up + timed + phase timing)
steps = max(int(r["Calls"]) for r in rows if "voice_audio_kernel" in r["Name"])
print(f"total kernel time {tot/1e6/steps:.3f} ms/step over {steps} steps")
for r in rows[:24]:
    print(f"{float(r['TotalDurationNs'])/tot*100:5.1f}%  {float(r['TotalDurationNs'])/1e6/steps:6.3f} ms/step  calls/step {int(r['Calls'])/steps:5.1f}  avg {float(r['AverageNs'])/1e3:8.1f} us  {r['Name'][:100]}")
PY
