"""Where a wave of stft_mfma_kernel spends its cycles: s_memtime stamps at the phase boundaries (diagnostic build
-DIAS_SM_STAMPS of the same source, linked into scripts/diag/_bin/libias_smstamps.so; the product library has none).
usage (GPU box): python scripts/diag/stft_stamps.py [loss|mel|raw]"""
import ctypes, os, subprocess, sys, collections
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CSRC = os.path.join(ROOT, "inverse-audio-synthesis_amd", "csrc")
BIN = os.path.join(ROOT, "scripts", "diag", "_bin")
os.makedirs(BIN, exist_ok=True)
so = os.path.join(BIN, "libias_smstamps.so")
objs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith("_kernels.o") and not f.startswith("stft_mfma")]
if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(os.path.join(CSRC, "stft_mfma_kernels.hip")):
    obj = os.path.join(BIN, "stft_mfma_stamps.o")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-fPIC", "-std=c++17", "--offload-arch=gfx950", "-Wno-unused-value", "-fno-slp-vectorize", "-mllvm", "-amdgpu-kernarg-preload-count=16",
                           "-Wno-pass-failed", "-DIAS_SM_STAMPS", "-c", os.path.join(CSRC, "stft_mfma_kernels.hip"), "-o", obj])
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-shared", "-fPIC", "--offload-arch=gfx950", obj] + objs + ["-o", so])
if "--build-only" in sys.argv:
    sys.exit(0)
os.environ["IAS_HIP_LIB"] = so
sys.path.insert(0, ROOT)
import torch
from inverse_audio_synthesis_amd import _lib
from inverse_audio_synthesis_amd.spectral import MelSpectrogramL1, STFTPlan, VALUE_POWER
lib = _lib.load()
lib.ias_stft_set_stamps.restype = ctypes.c_int
lib.ias_stft_set_stamps.argtypes = [ctypes.c_void_p]
dev = torch.device("cuda:0")
mode = sys.argv[1] if len(sys.argv) > 1 else "loss"
a = (torch.randn(128, 176400, generator=torch.Generator().manual_seed(0)) * 0.1).to(dev)
mel = MelSpectrogramL1().to(dev)
tm = mel.target(a).clone()
raw = STFTPlan(1024, None, 512).to(dev)
fn = {"loss": lambda: mel(a, target_mel=tm), "mel": lambda: mel.mel.plan.values(a, VALUE_POWER),
      "raw": lambda: raw.values(a, VALUE_POWER)}[mode]
for _ in range(3): fn()
torch.cuda.synchronize()
grid = 768
buf = torch.zeros(grid * 4 * 256, dtype=torch.int64, device=dev)
lib.ias_stft_set_stamps(ctypes.c_void_p(buf.data_ptr()))
fn(); torch.cuda.synchronize()
lib.ias_stft_set_stamps(None)
st = buf.cpu().view(grid, 4, 256)
names = {1: "group start / ticket / barrier B", 2: "stage 1 (load wait, window, 32 MFMA)", 3: "next-frame loads, twiddle 1",
         4: "stage 2a (16 MFMA)", 5: "twiddle 2 + radix 4 + Z->LDS", 6: "unpack (Z<-LDS, power)", 7: "P->LDS / epilogue",
         8: "loop", 9: "weights, target row, barrier A", 10: "mel tiles + loss"}
tot = collections.defaultdict(int); cnt = collections.defaultdict(int)
spans = []
for wg in range(0, grid, 7):
    for w in range(4):
        n = int(st[wg, w, 0])
        if n < 2: continue
        v = st[wg, w, 1:1 + n].tolist()
        ids = [x & 255 for x in v]; ts = [x >> 8 for x in v]
        for i in range(1, n):
            tot[(ids[i], w)] += ts[i] - ts[i - 1]; cnt[(ids[i], w)] += 1
        spans.append((ts[-1] - ts[0], ids.count(1)))
print(f"mode {mode}: s_memtime ticks per phase (phase = the interval ENDING at the stamp), by wave of the workgroup")
for i in sorted(names):
    row = "  ".join(f"w{w}: {tot[(i, w)] / max(cnt[(i, w)], 1):7.0f} x{cnt[(i, w)]:4d}" for w in range(4))
    print(f"{i:2d} {names[i]:40s} {row}")
fr = sum(s[1] for s in spans); tt = sum(s[0] for s in spans)
print(f"stamped span per frame: {tt / max(fr, 1):.0f} ticks over {fr} frames")
