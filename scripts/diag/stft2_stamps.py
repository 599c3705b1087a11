"""Where a wave of stft2_kernel spends its cycles: s_memtime stamps at the phase boundaries (diagnostic build
-DIAS_S2_STAMPS of csrc/spectral_kernels.hip, linked into scripts/diag/_bin/libias_s2stamps.so; the product library has
none).  usage (GPU box): python scripts/diag/stft2_stamps.py [loss|mel|raw|mr1024|mr2048]
Caveat (round 4): the stamped build of the MR-STFT variants SPILLS (scratch reloads + s_waitcnt vmcnt(0) between stamps 7 and 8,
which then absorb the latency of the next frame's prefetch): phase 8 reads 3-4 x too long there; the product build has no scratch."""
import ctypes, os, subprocess, sys, collections
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CSRC = os.path.join(ROOT, "inverse-audio-synthesis_amd", "csrc")
BIN = os.path.join(ROOT, "scripts", "diag", "_bin")
os.makedirs(BIN, exist_ok=True)
so = os.path.join(BIN, "libias_s2stamps.so")
objs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith("_kernels.o") and not f.startswith("spectral_kernels")]
if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(os.path.join(CSRC, "spectral_kernels.hip")):
    obj = os.path.join(BIN, "spectral_s2stamps.o")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-fPIC", "-std=c++17", "--offload-arch=gfx950", "-Wno-unused-value",
                           "-fno-slp-vectorize", "-mllvm", "-amdgpu-kernarg-preload-count=16", "-DIAS_S2_STAMPS", "-c", os.path.join(CSRC, "spectral_kernels.hip"), "-o", obj])
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-shared", "-fPIC", "--offload-arch=gfx950", obj] + objs + ["-o", so])
if "--build-only" in sys.argv:
    sys.exit(0)
os.environ["IAS_HIP_LIB"] = so
sys.path.insert(0, ROOT)
import torch
from inverse_audio_synthesis_amd import _lib
from inverse_audio_synthesis_amd.spectral import (MelSpectrogramL1, STFTPlan, VALUE_POWER, VALUE_MAG_CLAMPED, LOSS_MRSTFT)
lib = _lib.load()
lib.ias_stft2_set_stamps.restype = ctypes.c_int
lib.ias_stft2_set_stamps.argtypes = [ctypes.c_void_p]
dev = torch.device("cuda:0")
mode = sys.argv[1] if len(sys.argv) > 1 else "loss"
B = 64 if mode.startswith("mr") else 128
a = (torch.randn(B, 176400, generator=torch.Generator().manual_seed(0)) * 0.1).to(dev)
if mode.startswith("mr"):
    n, h, w = (1024, 120, 600) if mode == "mr1024" else (2048, 240, 1200)
    plan = STFTPlan(n, w, h).to(dev)
    tg = plan.values(a, VALUE_MAG_CLAMPED, 1e-8).clone()
    fn = lambda: plan.loss_sums(a, tg, VALUE_MAG_CLAMPED, LOSS_MRSTFT, 1e-8)
else:
    mel = MelSpectrogramL1().to(dev)
    tm = mel.target(a).clone()
    raw = STFTPlan(1024, None, 512).to(dev)
    fn = {"loss": lambda: mel(a, target_mel=tm), "mel": lambda: mel.mel.plan.values(a, VALUE_POWER),
          "raw": lambda: raw.values(a, VALUE_POWER)}[mode]
for _ in range(3): fn()
torch.cuda.synchronize()
grid, W = 512, 8
buf = torch.zeros(grid * W * 256, dtype=torch.int64, device=dev)
lib.ias_stft2_set_stamps(ctypes.c_void_p(buf.data_ptr()))
fn(); torch.cuda.synchronize()
lib.ias_stft2_set_stamps(None)
st = buf.cpu().view(grid, W, 256)
names = {1: "loop top (prev emit tail, lds sync, xc<-xn)", 2: "next-frame loads issued, window, dft8, tw1, 8 ds_write",
         3: "wait exchange-1 writes", 4: "8 ds_read + wait", 5: "dft8, tw2, 8 ds_write", 6: "wait exchange-2 writes",
         7: "8 ds_read + wait", 8: "dft8 (+combine), half-Z ds_write", 9: "wait half-Z writes", 10: "unpack: reads + power",
         11: "wait (Z reads done)", 12: "segment-major scatter (9 ds_write_b32)", 13: "wait scatter", 14: "segment sums (rows x (b32 + b64))",
         15: "U/D shift, loss / emit"}
tot = collections.defaultdict(int); cnt = collections.defaultdict(int)
span = 0; frames = 0
for wg in range(0, grid, 5):
    for w in range(W):
        n = int(st[wg, w, 0])
        if n < 2: continue
        v = st[wg, w, 1:1 + n].tolist()
        ids = [x & 255 for x in v]; ts = [x >> 8 for x in v]
        for i in range(1, n):
            tot[ids[i]] += ts[i] - ts[i - 1]; cnt[ids[i]] += 1
        span += ts[-1] - ts[0]; frames += ids.count(1) - 1 if ids[-1] == 1 else ids.count(1)
print(f"mode {mode}: s_memtime ticks (100 MHz: 1 tick = 10 ns = ~21 shader cycles) per phase = the interval ENDING at the stamp")
s = 0.0
for i in sorted(names):
    if cnt[i]:
        print(f"{i:2d} {names[i]:58s} {tot[i] / cnt[i]:8.1f}  x{cnt[i]}")
        s += tot[i] / cnt[i]
print(f"sum of phases {s:.1f} ticks per frame ({s * 10:.0f} ns); stamped span per frame {span / max(frames, 1):.1f} ticks over {frames} frames")
