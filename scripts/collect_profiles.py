"""Copies the judged summaries of one scripts/refresh_profiles.sh run from gpurun_out/<tag>/ into profiles/<prefix>_*.
usage: python scripts/collect_profiles.py <tag> <prefix>      (run in the repo, after the gpurun call has merged back)"""
import os, shutil, sys

tag, prefix = sys.argv[1], sys.argv[2]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src, dst = os.path.join(root, "gpurun_out", tag), os.path.join(root, "profiles")
names = ["bench_default.json", "bench_nopipeline.json", "bench_vicreg128.json", "bench_vicreg1024.json", "bench_gradstep.json",
         "kstats_default.csv", "kstats_nopipeline.csv", "kstats_vicreg128.csv", "kstats_vicreg1024.csv", "kstats_gradstep.csv",
         "pmc_voice.txt", "pmc_pqmf.txt", "pmc_vicreg.txt", "pmc_vicreg1024.txt", "kstats_pretrain.txt", "pmc_stft.txt", "trace_default.txt", "trace_gradstep.txt",
         "mfma_valu_overlap.txt", "mfma_valu_inwave.txt", "parity_vs_torch_seed0.json", "parity_vs_torch_seed1.json"]
for n in names:
    p = os.path.join(src, n)
    if os.path.exists(p) and os.path.getsize(p) > 0:
        shutil.copy(p, os.path.join(dst, f"{prefix}_{n}"))
        print("copied", n)
    else:
        print("MISSING", n)
c = os.path.join(src, "counters.json")
if os.path.exists(c):
    shutil.copy(c, os.path.join(dst, "counters.json"))
    shutil.copy(c, os.path.join(dst, f"{prefix}_counters.json"))
    print("copied counters.json")
t = os.path.join(src, "traffic.json")
if os.path.exists(t):
    shutil.copy(t, os.path.join(dst, "traffic.json"))
    shutil.copy(t, os.path.join(dst, f"{prefix}_traffic.json"))
    print("copied traffic.json")
