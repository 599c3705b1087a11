#!/bin/bash
# kernel timeline of one replayed configs[4] gradient step: bash scripts/diag/trace_gradstep.sh <tag> [ENV=.. ...]
tag=$1; shift
R=$GRAFT_REPO_ROOT
for kv in "$@"; do export "$kv"; done
O=$R/gpurun_out/trace_gs_$tag
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $O/p -o out --output-format csv -- python3 $R/bench.py --workload gradstep --no-cpu-baseline --no-legs --steps 20 --warmup 3 > $O/log.txt 2>&1
python3 - <<PY
import csv, glob, re
f = glob.glob("$O/p/**/out_kernel_trace.csv", recursive=True)[0]
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r["Queue_Id"]) for r in csv.DictReader(open(f)))
ri = [i for i, x in enumerate(ev) if "voice_audio_kernel" in x[2]]
i0, i1 = ri[-3], ri[-2]                      # one whole step late in the run (graph replays)
t0 = ev[i0][0]
for s, e, n, q in ev[i0:i1]:
    n = re.sub(r"\(.*", "", n).replace("void ", "")[:44]
    print(f"{(s-t0)/1e3:8.1f} -> {(e-t0)/1e3:8.1f} ({(e-s)/1e3:6.1f}) q{q:>2s} {n}")
print(f"step period {(ev[i1][0]-t0)/1e3:.1f} us")
PY
rm -rf $O/p
