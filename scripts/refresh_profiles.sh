#!/bin/bash
# Round-end measurement set, run on the GPU box:  bash scripts/refresh_profiles.sh <tag>
# Writes into gpurun_out/<tag>/ :
#   bench_{default,nopipeline,vicreg128,vicreg1024,gradstep}.json    the bench lines
#   kstats_{default,nopipeline,vicreg128,vicreg1024,gradstep}.csv     rocprofv3 --kernel-trace --stats of the same commands
#   traffic.json                                                      FETCH_SIZE / WRITE_SIZE PMC passes (separate runs) over
#                                                                     scripts/diag/time_stages.py, per kernel and launch
#   pmc_voice.txt / pmc_pqmf.txt / pmc_vicreg.txt / pmc_vicreg1024.txt / kstats_pretrain.txt   SQ counter summaries of the named kernels
#   counters.json                                                     the SQ counters bench.py imports (scripts/make_counters.py over pmc_{voice,stft,pqmf}.txt)
# Copy what is to be judged into profiles/ afterwards (scripts/collect_profiles.py <tag> <prefix>).
tag=${1:-rXX}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$tag
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
step() { echo "[refresh_profiles] $*" >> $O/progress.log; echo "[refresh_profiles] $*"; }

step bench default; python3 $R/bench.py > $O/bench_default.json 2> $O/bench_default.err
step bench nopipeline; python3 $R/bench.py --no-pipeline --no-cpu-baseline > $O/bench_nopipeline.json 2> $O/bench_nopipeline.err
step bench vicreg; python3 $R/bench.py --workload vicreg > $O/bench_vicreg128.json 2> $O/bench_vicreg128.err
python3 $R/bench.py --workload vicreg --batch 1024 --no-cpu-baseline > $O/bench_vicreg1024.json 2> $O/bench_vicreg1024.err
step bench gradstep; python3 $R/bench.py --workload gradstep --steps 20 --warmup 3 --no-cpu-baseline > $O/bench_gradstep.json 2> $O/bench_gradstep.err

kst() { # name, bench args...
  local name=$1; shift
  step kstats $name
  rocprofv3 --kernel-trace --stats -d $O/prof_$name -o out --output-format csv -- python3 $R/bench.py "$@" > $O/prof_$name.log 2>&1
  f=$(find $O/prof_$name -name out_kernel_stats.csv | head -1); [ -n "$f" ] && cp $f $O/kstats_$name.csv
  rm -rf $O/prof_$name          # the raw kernel trace is tens of MB; gpurun merges at most 64 MiB back
}
kst default --no-cpu-baseline --no-legs
kst nopipeline --no-pipeline --no-graph --no-cpu-baseline
kst vicreg128 --workload vicreg --no-cpu-baseline --no-graph --steps 20 --warmup 3
kst vicreg1024 --workload vicreg --batch 1024 --no-cpu-baseline --no-graph --steps 20 --warmup 3
kst gradstep --workload gradstep --steps 5 --warmup 2 --no-cpu-baseline --no-graph

step pmc traffic
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/pmc_fetch -o out --output-format csv -- python3 $R/scripts/diag/time_stages.py > $O/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O/pmc_write -o out --output-format csv -- python3 $R/scripts/diag/time_stages.py > $O/pmc_write.log 2>&1
python3 - <<PY
import csv, glob, json, collections
O = "$O"
def avg(counter, pat):
    agg = collections.defaultdict(list)
    for f in glob.glob(pat, recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                agg[r["Kernel_Name"].split("(")[0]].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in agg.items()}
fe = avg("FETCH_SIZE", O + "/pmc_fetch/**/out_counter_collection.csv")
wr = avg("WRITE_SIZE", O + "/pmc_write/**/out_counter_collection.csv")
out = {"_method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in two separate passes (kernel-trace only) over scripts/diag/time_stages.py (B=128 x 176400), scripts/refresh_profiles.sh; counters in KB; hbm_bytes_per_launch = (2*FETCH_SIZE + WRITE_SIZE)*1024 (gfx950 FETCH_SIZE reports 1/2 of wide coalesced reads, MI355X_MICROARCH.md section HBM; the factor is calibrated for 16-byte accesses only)", "_round": "$tag"}
for k in sorted(set(fe) | set(wr)):
    if k.startswith("void at::") or k.startswith("__amd"):
        continue
    out[k] = {"FETCH_SIZE_KB_avg": round(fe.get(k, 0.0), 1), "WRITE_SIZE_KB_avg": round(wr.get(k, 0.0), 1),
              "hbm_bytes_per_launch": int((2 * fe.get(k, 0.0) + wr.get(k, 0.0)) * 1024)}
json.dump(out, open(O + "/traffic.json", "w"), indent=1)
PY
rm -rf $O/pmc_fetch $O/pmc_write

step pmc voice; bash $R/scripts/diag/pmc_voice.sh $tag > /dev/null 2>&1; cp $R/gpurun_out/pmcv_$tag/summary.txt $O/pmc_voice.txt
step pmc pqmf; bash $R/scripts/diag/pmc_pqmf.sh $tag N=3 > /dev/null 2>&1; cp $R/gpurun_out/pmcq_$tag/summary.txt $O/pmc_pqmf.txt
step pmc vicreg; bash $R/scripts/diag/pmc_vicreg.sh $tag 128 > /dev/null 2>&1; cp $R/gpurun_out/pmcg_$tag/summary.txt $O/pmc_vicreg.txt
step pmc vicreg1024; bash $R/scripts/diag/pmc_vicreg.sh ${tag}_1024 1024 > /dev/null 2>&1; cp $R/gpurun_out/pmcg_${tag}_1024/summary.txt $O/pmc_vicreg1024.txt
# the Gram kernels' FETCH_SIZE / WRITE_SIZE (their own passes inside pmc_vicreg.sh) -> traffic.json, for the vicreg bench lines
python3 - <<PY
import json, re
O = "$O"
def parse(path):
    out, cur = {}, None
    try:
        for line in open(path):
            if line[:1] not in (" ", "\n") and "calls" not in line:
                cur = line.split("(")[0].strip(); out.setdefault(cur, {})
            else:
                m = re.match(r"^\s+(\w+)\s+([0-9.]+)", line)
                if m and cur: out[cur][m.group(1)] = float(m.group(2))
    except OSError:
        pass
    return out
t = json.load(open(O + "/traffic.json"))
for path, kname in ((O + "/pmc_vicreg.txt", "vicreg_gram_pair_kernel"), (O + "/pmc_vicreg1024.txt", "vicreg_gram256_kernel")):
    d = parse(path).get(kname, {})
    if "FETCH_SIZE" in d and "WRITE_SIZE" in d:
        t[kname] = {"FETCH_SIZE_KB_avg": round(d["FETCH_SIZE"], 1), "WRITE_SIZE_KB_avg": round(d["WRITE_SIZE"], 1),
                    "hbm_bytes_per_launch": int((2 * d["FETCH_SIZE"] + d["WRITE_SIZE"]) * 1024),
                    "_note": "bytes beyond L2 (FETCH_SIZE counts L2 misses; part of them are served by the Infinity Cache)"}
json.dump(t, open(O + "/traffic.json", "w"), indent=1)
PY
step kstats pretrain; bash $R/scripts/diag/kstats_pretrain.sh $tag > $O/kstats_pretrain.txt 2>&1
step pmc stft; bash $R/scripts/diag/pmc_stft.sh $tag PARTS=loss > /dev/null 2>&1; cp $R/gpurun_out/pmcs_$tag/summary.txt $O/pmc_stft.txt
step counters; python3 $R/scripts/make_counters.py $O $tag $O/counters.json > /dev/null 2>&1
step trace; bash $R/scripts/diag/trace_bench.sh $tag > $O/trace_default.txt 2>&1
step trace gradstep; bash $R/scripts/diag/trace_gradstep.sh $tag > $O/trace_gradstep.txt 2>&1; rm -rf $R/gpurun_out/trace_gs_$tag
step microbench; $R/scripts/diag/_bin/mfma_valu_overlap > $O/mfma_valu_overlap.txt 2>&1; $R/scripts/diag/_bin/mfma_valu_inwave > $O/mfma_valu_inwave.txt 2>&1
step parity; IAS_PARITY_OUT=$O python3 -m pytest $R/tests/test_voice_gpu.py -q -k headline_size > $O/parity_test.log 2>&1
# raw counter / trace directories of the helper scripts: the summaries above are what is kept
rm -rf $R/gpurun_out/pmcv_$tag $R/gpurun_out/pmcq_$tag $R/gpurun_out/pmcg_$tag $R/gpurun_out/pmcg_${tag}_1024 $R/gpurun_out/pmcs_$tag \
       $R/gpurun_out/kstats_pt_$tag/prof $R/gpurun_out/trace_$tag
step done
head -c 600 $O/bench_default.json; echo
