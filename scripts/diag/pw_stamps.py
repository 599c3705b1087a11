"""Where a wave of pw_apply_kernel spends its time: s_memrealtime stamps (diagnostic build -DIAS_PW_STAMPS of
csrc/pointwise_kernels.hip linked into scripts/diag/_bin/libias_pwstamps.so; the product library has none).
usage (GPU box): python scripts/diag/pw_stamps.py"""
import ctypes, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CSRC = os.path.join(ROOT, "inverse-audio-synthesis_amd", "csrc")
BIN = os.path.join(ROOT, "scripts", "diag", "_bin")
os.makedirs(BIN, exist_ok=True)
so = os.path.join(BIN, "libias_pwstamps.so")
objs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith("_kernels.o") and not f.startswith("pointwise_kernels")]
obj = os.path.join(BIN, "pointwise_pwstamps.o")
subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-fPIC", "-std=c++17", "--offload-arch=gfx950", "-Wno-unused-value", "-fno-slp-vectorize", "-mllvm", "-amdgpu-kernarg-preload-count=16",
                       "-DIAS_PW_STAMPS", "-c", os.path.join(CSRC, "pointwise_kernels.hip"), "-o", obj])
subprocess.check_call(["/opt/rocm/bin/hipcc", "-shared", "-fPIC", "--offload-arch=gfx950", obj] + objs + ["-o", so])
if "--build-only" in sys.argv:
    sys.exit(0)
os.environ["IAS_HIP_LIB"] = so
sys.path.insert(0, ROOT)
import torch
from inverse_audio_synthesis_amd import _lib
lib = _lib.load()
lib.ias_pw_set_stamps.restype = ctypes.c_int
lib.ias_pw_set_stamps.argtypes = [ctypes.c_void_p]
dev = torch.device("cuda:0")
B = 128
st = _lib.stream()
for ci, co, hw in [(240, 40, 240), (40, 240, 240), (96, 40, 240), (144, 48, 240), (24, 88, 930), (16, 72, 3720)]:
    x = torch.randn(B, ci, hw, device=dev)
    w = torch.randn(co, ci, device=dev)
    y = torch.empty(B, co, hw, device=dev)
    big = torch.empty(64 << 20, device=dev)                     # 256 MB: pushes x out of the Infinity Cache between calls
    fn = lambda: lib.ias_pwconv_forward(_lib.ptr(x), _lib.ptr(w), _lib.ptr(y), B, ci, co, hw, st)
    for _ in range(2):
        fn()
    big.zero_()
    torch.cuda.synchronize()
    nw = 1024 * 4
    buf = torch.zeros(nw * 8, dtype=torch.int64, device=dev)
    lib.ias_pw_set_stamps(ctypes.c_void_p(buf.data_ptr()))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); fn(); e1.record()
    torch.cuda.synchronize()
    lib.ias_pw_set_stamps(None)
    s = buf.cpu().view(nw, 8)
    live = s[:, 0] > 0
    s = s[live].double()
    t0 = s[:, 0].min()
    rel = (s[:, :5] - t0) / 100.0                                # us (100 MHz counter)
    def med(v): return float(v.median())
    print(f"{ci:4d} -> {co:4d} @ {hw:5d}: kernel {e0.elapsed_time(e1) * 1e3:6.1f} us, waves {int(live.sum())}; per wave (median us): "
          f"start {med(rel[:, 0]):5.1f}  weight staged +{med(rel[:, 1] - rel[:, 0]):4.1f}  first loads issued +{med(rel[:, 2] - rel[:, 1]):4.1f}  "
          f"k-loop +{med(rel[:, 3] - rel[:, 2]):5.1f}  stores +{med(rel[:, 4] - rel[:, 3]):4.1f}  end at {med(rel[:, 4]):5.1f} (max {float(rel[:, 4].max()):5.1f})")
