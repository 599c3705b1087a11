"""Per-step timing of vicreg_gram_pair_kernel from in-kernel s_memtime stamps (diagnostic build libias_gpstamps.so)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ["IAS_HIP_LIB"] = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_bin", os.environ.get("GP_LIB", "libias_gpstamps.so"))
import torch
from inverse_audio_synthesis_amd import _lib
lib = _lib.load()
dev = torch.device("cuda:0")
B, D = 128, 8192
x = torch.randn(B, D, device=dev); y = torch.randn(B, D, device=dev)
need = int(lib.ias_vicreg_workspace_bytes(B, D)); ws = torch.empty(need, dtype=torch.uint8, device=dev); o4 = torch.empty(4, device=dev)
def stage(k):
    _lib.check(lib.ias_vicreg_stage(k, _lib.ptr(x), _lib.ptr(y), _lib.ptr(o4), _lib.ptr(ws), need, B, D, B, 25.0, 25.0, 1.0, _lib.stream()), "stage")
stage(-1)
for _ in range(3): stage(1)
torch.cuda.synchronize()
st = torch.zeros((512, 8, 32), dtype=torch.int64, device=dev)
raw = ctypes.CDLL(os.environ["IAS_HIP_LIB"])
assert raw.ias_vicreg_debug_set_stamps(ctypes.c_void_p(st.data_ptr())) == 0
stage(1); torch.cuda.synchronize()
s = st.cpu().double()
s = s[s[:, 0, 0] > 0]
full = s[(s[:, 0, 21] > 0)]          # items with 10 steps
print(f"{s.shape[0]} workgroups, {full.shape[0]} ten-step items")
print("HW_ID simd field per wave (first item):", [int(full[0, w, 31].item()) for w in range(8)])
for w in range(8):
    f = full[:, w]
    setup = (f[:, 1] - f[:, 0]).mean().item()
    waits = [(f[:, 2 + 2 * k] - (f[:, 1] if k == 0 else f[:, 1 + 2 * k])).mean().item() for k in range(10)]
    comps = [(f[:, 3 + 2 * k] - f[:, 2 + 2 * k]).mean().item() for k in range(10)]
    print(f"wave {w}: setup {setup:6.0f}; steady wait+barrier {sum(waits[3:]) / 7:6.0f}, MFMA stream {sum(comps[3:]) / 7:6.0f}; item total {(f[:, 30] - f[:, 0]).mean().item():7.0f}")

for w in (0, 4):
    f = full[:, w]
    waits = [(f[:, 2 + 2 * k] - (f[:, 1] if k == 0 else f[:, 1 + 2 * k])).mean().item() for k in range(10)]
    comps = [(f[:, 3 + 2 * k] - f[:, 2 + 2 * k]).mean().item() for k in range(10)]
    print(f"wave {w} per step: gap before " + " ".join(f"{x:5.0f}" for x in waits) + " | stream " + " ".join(f"{x:5.0f}" for x in comps))
