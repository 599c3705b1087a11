#!/bin/bash
# kernel timeline of the headline bench: bash scripts/diag/trace_bench.sh <tag> [bench args...]
tag=$1; shift
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/trace_$tag
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $O/p -o out --output-format csv -- python3 $R/bench.py --no-cpu-baseline --no-legs --steps 40 --warmup 3 --replays 3 "$@" > $O/log.txt 2>&1
python3 - <<PY
import csv, glob
f = glob.glob("$O/p/**/out_kernel_trace.csv", recursive=True)[0]
ev = []
for r in csv.DictReader(open(f)):
    n = r["Kernel_Name"]
    s = "render" if "voice_audio" in n else "stft" if ("stft_kernel" in n or "stft2_kernel" in n) else "pqmf" if "pqmf_analysis" in n else "env" if "voice_env" in n else "lfo" if "voice_lfo" in n else "modmix" if "modmix" in n else "reduce" if "reduce_partials" in n else "fill" if "fillBuffer" in n else None
    if s: ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), s, r["Queue_Id"]))
ev.sort()
# a window inside a replay of the captured K-step schedule: the render at 45 % of all render launches
ri = [i for i, x in enumerate(ev) if x[2] == "render"]
i0 = ri[int(len(ri) * 0.45)]
t0 = ev[i0][0]
for s, e, n, q in ev[i0:i0 + 40]:
    print(f"{(s-t0)/1e3:9.1f} -> {(e-t0)/1e3:9.1f} us  ({(e-s)/1e3:6.1f})  {n:8s} q{q}")
# per-kernel in-step durations over the steady part of that replay (100 launches around the window)
import collections
dur = collections.defaultdict(list)
for s, e, n, q in ev[max(0, i0 - 150):i0 + 150]:
    dur[n].append((e - s) / 1e3)
print("in-step duration per launch (us), median over the launches around the window:")
for n in ("render", "stft", "pqmf", "env", "lfo", "modmix", "reduce", "fill"):
    if dur[n]:
        d = sorted(dur[n]); print(f"   {n:8s} median {d[len(d)//2]:7.1f}  min {d[0]:7.1f}  max {d[-1]:7.1f}  n={len(d)}")
rs = [ev[i][0] for i in ri]
k = int(len(rs) * 0.45)
d = [(b - a) / 1e3 for a, b in zip(rs[k - 15:k + 15], rs[k - 14:k + 16])]
print("render-to-render period (us): median", sorted(d)[len(d)//2])
PY
