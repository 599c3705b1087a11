#!/bin/bash
# headline bench with the PQMF on the matrix cores (default) and on the VALU kernels
R=$GRAFT_REPO_ROOT
cd $R
python3 bench.py --no-cpu-baseline > gpurun_out/bench_ab_mfma.json 2> gpurun_out/bench_ab_mfma.err
IAS_PQMF_VALU=1 python3 bench.py --no-cpu-baseline > gpurun_out/bench_ab_valu.json 2> gpurun_out/bench_ab_valu.err
IAS_PQM_TILED=1 python3 bench.py --no-cpu-baseline > gpurun_out/bench_ab_tiled.json 2> gpurun_out/bench_ab_tiled.err
python3 - <<PY
import json
for v in ("mfma", "valu", "tiled"):
    try:
        d = json.load(open(f"gpurun_out/bench_ab_{v}.json"))
        print(v, d["ms_per_step"], d.get("ms_per_step_min"), d["value"])
    except Exception as ex:
        print(v, "failed", ex)
PY
