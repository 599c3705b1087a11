"""Stem convolution kernels (forward, weight gradient) alone at the training step's shape [128,3,240,245], with the caches
flushed by a 1 GB fill between calls (the step's operands come from HBM).  LIB=diag + IAS_STEM_GW_GATHER=1: the
gather form of the weight gradient."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from inverse_audio_synthesis_amd import _lib
if os.environ.get("LIB") == "diag":
    _lib.use_library(_lib.DIAG_LIB_PATH)
lib = _lib.load()
dev = torch.device("cuda:0")
B, H, W = 128, 240, 245
Ho, Wo = (H + 1) // 2, (W + 1) // 2
x = torch.randn(B, 3, H, W, device=dev); g = torch.randn(B, 16, Ho, Wo, device=dev)
w = torch.randn(16, 3, 3, 3, device=dev); out = torch.empty(B, 16, Ho, Wo, device=dev)
gw = torch.empty(16, 3, 3, 3, device=dev)
scratch = torch.empty(int(lib.ias_stem_weight_scratch(B)), device=dev)
flush = torch.empty(256 * 1024 * 1024, device=dev)
def timeit(fn, n=10):
    ts = []
    for _ in range(n):
        flush.fill_(1.0)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort()
    return ts[len(ts) // 2], ts[0]
fw = lambda: _lib.check(lib.ias_stem_forward(_lib.ptr(x), _lib.ptr(w), _lib.ptr(out), B, H, W, _lib.stream()), "fwd")
bw = lambda: _lib.check(lib.ias_stem_backward_weight(_lib.ptr(x), _lib.ptr(g), _lib.ptr(gw), _lib.ptr(scratch), B, H, W, _lib.stream()), "gw")
for name, fn in (("stem forward", fw), ("stem weight gradient (+ reduce)", bw)):
    fn(); torch.cuda.synchronize()
    med, mn = timeit(fn)
    print(f"{name}: median {med:.1f} us  min {mn:.1f} us  (cold caches; lib={os.environ.get('LIB', 'product')} gather={os.environ.get('IAS_STEM_GW_GATHER', '0')})")
ref = torch.nn.grad.conv2d_weight(x, w.shape, g, stride=2, padding=1)
print("gw max rel err vs torch:", ((gw - ref).abs().max() / ref.abs().max()).item())
