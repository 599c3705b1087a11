#!/bin/bash
# every kernel of ONE replayed pretraining step in launch order: duration, grid, workgroup, LDS (rocprofv3 kernel trace)
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/trace_ptl
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
GRAPH=1 STEPS=3 rocprofv3 --kernel-trace -d $O/p -o out --output-format csv -- python3 $R/scripts/diag/time_pretrain_step.py > $O/log.txt 2>&1
python3 - <<PY > $O/list.txt
import csv, glob
f = glob.glob("$O/p/**/out_kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "lars_update" in r["Kernel_Name"]]
step = rows[idx[-2] + 1: idx[-1] + 1]
t0 = int(step[0]["Start_Timestamp"])
for r in step:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    grid = "x".join(str(r.get(k, "?")) for k in ("Grid_Size_X", "Grid_Size_Y", "Grid_Size_Z"))
    wg = r.get("Workgroup_Size_X", "?")
    print(f"{(s - t0) / 1e3:9.1f} {(e - s) / 1e3:8.1f} us  grid {grid:>16s} wg {wg:>5s} lds {r.get('LDS_Block_Size', '?'):>7s}  {r['Kernel_Name'][:90]}")
PY
rm -rf $O/p
wc -l $O/list.txt
