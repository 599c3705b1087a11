#!/bin/bash
# A/B of voice_kernels builds (scripts/diag/build_variants.sh voice_kernels ...) on one box: isolated render time and the
# headline step for each:  bash scripts/diag/run_voice_variants.sh base ilp ...
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
out=gpurun_out/voice_variants.txt; : > $out
for v in "$@"; do
  lib=$PWD/scripts/diag/_bin/libias_$v.so
  IAS_HIP_LIB=$lib python3 scripts/diag/time_voice.py >> $out 2>&1
  IAS_HIP_LIB=$lib python3 bench.py --steps 50 --no-legs --no-cpu-baseline > gpurun_out/bench_var_$v.json 2>> gpurun_out/bench_var.err
  python3 - $v >> $out <<PY
import json, sys
j = json.load(open(f"gpurun_out/bench_var_{sys.argv[1]}.json"))
print("   step", j["ms_per_step"], "min", j["ms_per_step_min"], {k: (v["isolated_avg_us"], v["in_step_avg_us"]) for k, v in j["roofline"]["kernels"].items()})
PY
done
cat $out
