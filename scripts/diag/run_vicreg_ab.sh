cd $GRAFT_REPO_ROOT
for i in 1 2 3; do
for lib in "" scripts/diag/_bin/libias_vicold.so; do
  if [ -n "$lib" ]; then export IAS_HIP_LIB=$PWD/$lib; else unset IAS_HIP_LIB; fi
  python bench.py --workload vicreg --no-cpu-baseline --steps 20 > gpurun_out/bench_v128.json 2>/dev/null
  python -c "import json,os; j=json.load(open('gpurun_out/bench_v128.json')); print(os.environ.get('IAS_HIP_LIB','new')[-14:], j['ms_per_step'], j['roofline']['avg_launch_ms'], j['roofline']['frac'])"
done; done
