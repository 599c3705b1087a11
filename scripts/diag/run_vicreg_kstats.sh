#!/bin/bash
# rocprofv3 kernel stats of bench.py --workload vicreg at B = 128 and 1024 (eager, 20 steps) -> gpurun_out/vk/kstats_<B>.txt
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/vk; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
for b in 128 1024; do
  rocprofv3 --kernel-trace --stats -d $O/p$b -o out --output-format csv -- python3 $R/bench.py --workload vicreg --batch $b --no-cpu-baseline --no-graph --steps 20 --warmup 3 > $O/p$b.log 2>&1
  f=$(find $O/p$b -name out_kernel_stats.csv | head -1)
  python3 - "$f" > $O/kstats_$b.txt <<'PY'
import csv, sys
for r in list(csv.reader(open(sys.argv[1])))[1:14]:
    print(f"{r[0][:60]:60s} calls {r[1]:>5s} avg {float(r[3])/1000:8.1f} us  min {float(r[5])/1000:8.1f}")
PY
  rm -rf $O/p$b
  cat $O/kstats_$b.txt
done
