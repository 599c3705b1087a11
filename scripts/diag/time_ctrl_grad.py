"""voice control-rate backward: one launch (SPLIT=0) vs the three-launch form -- developer tool."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from inverse_audio_synthesis_amd import _lib
from inverse_audio_synthesis_amd import voice_spec as S
lib = _lib.load()
dev = torch.device("cuda:0")
B, Tc = 64, 1764
g = torch.Generator().manual_seed(0)
p = torch.rand(B, 78, generator=g).to(dev)
g_ctrl = torch.randn(B, 5, Tc, generator=g).to(dev)
g_scal = torch.randn(B, 12, generator=g, dtype=torch.float64).to(dev)
out = torch.empty(B, 78, device=dev)
SPLIT = os.environ.get("SPLIT", "1") != "0"
ws = torch.empty(int(lib.ias_voice_control_backward_ws_bytes(B, Tc)), dtype=torch.uint8, device=dev)
def run():
    if SPLIT:
        st = lib.ias_voice_control_backward_ws(_lib.ptr(p), _lib.ptr(g_ctrl), _lib.ptr(g_scal), _lib.ptr(out), _lib.ptr(ws), ws.numel(), B, Tc, 441, _lib.stream())
    else:
        st = lib.ias_voice_control_backward(_lib.ptr(p), _lib.ptr(g_ctrl), _lib.ptr(g_scal), _lib.ptr(out), B, Tc, 441, _lib.stream())
    assert st == 0, st
for _ in range(3): run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): run()
e1.record(); torch.cuda.synchronize()
print("split" if SPLIT else "one launch", f"{e0.elapsed_time(e1) / 20 * 1000:.1f} us", float(out.abs().sum()))
