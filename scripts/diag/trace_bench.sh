#!/bin/bash
# kernel timeline of the headline bench: bash scripts/diag/trace_bench.sh <tag> [bench args...]
tag=$1; shift
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/trace_$tag
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $O/p -o out --output-format csv -- python3 $R/bench.py --no-cpu-baseline --steps 40 --warmup 3 --replays 3 "$@" > $O/log.txt 2>&1
python3 - <<PY
import csv, glob
f = glob.glob("$O/p/**/out_kernel_trace.csv", recursive=True)[0]
ev = []
for r in csv.DictReader(open(f)):
    n = r["Kernel_Name"]
    s = "render" if "voice_audio" in n else "stft" if "stft_kernel" in n else "pqmf" if "pqmf_analysis" in n else "env" if "voice_env" in n else "lfo" if "voice_lfo" in n else "modmix" if "modmix" in n else "reduce" if "reduce_partials" in n else "fill" if "fillBuffer" in n else None
    if s: ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), s, r["Queue_Id"]))
ev.sort()
i0 = int(len(ev) * 0.8)
t0 = ev[i0][0]
for s, e, n, q in ev[i0:i0 + 36]:
    print(f"{(s-t0)/1e3:9.1f} -> {(e-t0)/1e3:9.1f} us  ({(e-s)/1e3:6.1f})  {n:8s} q{q}")
rs = [s for s, e, n, q in ev if n == "render"]
d = [(b - a) / 1e3 for a, b in zip(rs[len(rs)//2:], rs[len(rs)//2+1:])]
print("render-to-render period (us): median", sorted(d)[len(d)//2])
PY
