#!/bin/bash
# rocprofv3 kernel stats of one VICReg pretraining step (scripts/diag/time_pretrain_step.py).  usage: bash scripts/diag/kstats_pretrain.sh <tag>
tag=${1:-x}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/kstats_pt_$tag
mkdir -p $O
python3 $R/scripts/diag/time_pretrain_step.py > $O/time.log 2>&1
GRAPH=1 python3 $R/scripts/diag/time_pretrain_step.py > $O/time_graph.log 2>&1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/prof -o out --output-format csv -- python3 $R/scripts/diag/time_pretrain_step.py > $O/prof.log 2>&1
python3 - <<PY
import csv, glob
rows = []
for f in glob.glob("$O/prof/**/out_kernel_stats.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(open("$O/time.log").read().strip().splitlines()[-1])
print(open("$O/time_graph.log").read().strip().splitlines()[-1])
print(f"total kernel time {tot/1e6:.1f} ms over 10 steps (5 warm-up + 5 timed)")
import os, re
fam = {}
def family(n):
    for pat, f in (("Cijk_", "library GEMM"), ("bn_", "BatchNorm"), ("dw_tile|dw_", "depthwise"), ("stem_|conv2x2", "stem / head patches"),
                   ("pw_", "thin 1x1"), ("se_", "squeeze-excitation"), ("lars_", "LARS"), ("at::native|rocclr", "torch / runtime"),
                   ("voice_|pqmf_", "render + PQMF"), ("vicreg_", "VICReg loss")):
        if re.search(pat, n):
            return f
    return "other"
for r in rows:
    fam[family(r["Name"])] = fam.get(family(r["Name"]), 0.0) + float(r["TotalDurationNs"])
print("by family, us per step: " + ", ".join(f"{k} {v/1e4:.0f}" for k, v in sorted(fam.items(), key=lambda kv: -kv[1])))
for r in rows[:int(os.environ.get("KSTATS_ROWS", "28"))]:
    print(f"{float(r['TotalDurationNs'])/tot*100:5.1f}%  calls {r['Calls']:>6s}  avg {float(r['AverageNs'])/1e3:9.1f} us  {r['Name'][:110]}")
PY
