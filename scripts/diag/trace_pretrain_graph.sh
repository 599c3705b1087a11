#!/bin/bash
# gaps between the kernels of one replayed pretraining step (rocprofv3 kernel trace)
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/trace_ptg
mkdir -p $O
for i in 1 2 3; do GRAPH=1 STEPS=10 python3 $R/scripts/diag/time_pretrain_step.py 2>&1 | tail -1; done
cd /tmp && export TMPDIR=/tmp
GRAPH=1 STEPS=4 rocprofv3 --kernel-trace -d $O/p -o out --output-format csv -- python3 $R/scripts/diag/time_pretrain_step.py > $O/log.txt 2>&1
python3 - <<PY
import csv, glob
f = glob.glob("$O/p/**/out_kernel_trace.csv", recursive=True)[0]
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:50]) for r in csv.DictReader(open(f)))
# last replayed step: find the last lars_update and the one before
idx = [i for i, e in enumerate(ev) if "lars_update" in e[2]]
a, b = idx[-2] + 1, idx[-1] + 1
step = ev[a:b]
busy = sum(e - s for s, e, _ in step)
span = step[-1][1] - step[0][0]
gaps = sorted(((step[i + 1][0] - step[i][1]) / 1e3, step[i][2], step[i + 1][2]) for i in range(len(step) - 1))
print(f"kernels {len(step)}  span {span/1e6:.2f} ms  busy {busy/1e6:.2f} ms")
import statistics
g = [x[0] for x in gaps]
print("gap us: median", round(statistics.median(g), 1), "mean", round(sum(g) / len(g), 1), "p90", round(g[int(len(g) * 0.9)], 1), "max", round(g[-1], 1))
for x in gaps[-8:]:
    print("  ", round(x[0], 1), "us between", x[1], "->", x[2])
PY
