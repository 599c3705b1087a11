#!/bin/bash
# Round-end measurement set, run on the GPU box:  bash scripts/refresh_profiles.sh <tag>
# Writes into gpurun_out/<tag>/ : bench JSON lines (default + --no-pipeline), rocprofv3 kernel stats of both
# schedules, and the FETCH_SIZE / WRITE_SIZE PMC passes (separate runs, kernel-trace only) over time_stages.py.
tag=${1:-rXX}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$tag
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py > $O/bench_default.json 2> $O/bench_default.err
python3 $R/bench.py --no-pipeline --no-cpu-baseline > $O/bench_nopipeline.json 2> $O/bench_nopipeline.err
rocprofv3 --kernel-trace --stats -d $O/prof_default -o out --output-format csv -- python3 $R/bench.py --no-cpu-baseline > $O/prof_default.log 2>&1
rocprofv3 --kernel-trace --stats -d $O/prof_nopipeline -o out --output-format csv -- python3 $R/bench.py --no-pipeline --no-graph --no-cpu-baseline > $O/prof_nopipeline.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/pmc_fetch -o out --output-format csv -- python3 $R/scripts/diag/time_stages.py > $O/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O/pmc_write -o out --output-format csv -- python3 $R/scripts/diag/time_stages.py > $O/pmc_write.log 2>&1
find $O -name "out_kernel_stats.csv" -o -name "out_counter_collection.csv" | while read f; do d=$(basename $(dirname $(dirname $f))); cp $f $O/${d}_$(basename $f); done
python3 - <<PY
import csv, glob, json, collections
O = "$O"
def avg(counter, path):
    agg = collections.defaultdict(list)
    for f in glob.glob(path):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                agg[r["Kernel_Name"].split("(")[0]].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in agg.items()}
fe = avg("FETCH_SIZE", O + "/pmc_fetch_out_counter_collection.csv")
wr = avg("WRITE_SIZE", O + "/pmc_write_out_counter_collection.csv")
out = {"_method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in two separate passes over scripts/diag/time_stages.py (B=128 x 176400); counters in KB; hbm_bytes_per_launch = (2*FETCH_SIZE + WRITE_SIZE)*1024 (gfx950 FETCH_SIZE reports 1/2 of wide coalesced reads, MI355X_MICROARCH.md section HBM; the factor is calibrated for 16-byte accesses only)", "_round": "$tag"}
for k in sorted(set(fe) | set(wr)):
    if k.startswith("void at::") or k.startswith("__amd"):
        continue
    out[k] = {"FETCH_SIZE_KB_avg": round(fe.get(k, 0.0), 1), "WRITE_SIZE_KB_avg": round(wr.get(k, 0.0), 1),
              "hbm_bytes_per_launch": int((2 * fe.get(k, 0.0) + wr.get(k, 0.0)) * 1024)}
json.dump(out, open(O + "/traffic.json", "w"), indent=1)
print(open(O + "/bench_default.json").read()[:400])
PY
