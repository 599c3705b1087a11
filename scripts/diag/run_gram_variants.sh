for v in gpnoload gpnomfma; do IAS_HIP_LIB=$PWD/scripts/diag/_bin/libias_$v.so timeout -k 10 100 python bench.py --workload vicreg --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null > gpurun_out/v_$v.json; python -c "
import json; d=json.load(open('gpurun_out/v_$v.json')); print('$v', d['roofline']['avg_launch_ms'])"; done
