"""PQMF analysis kernel alone: K launches captured in one hipGraph, HIP-event timed -- developer tool."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from inverse_audio_synthesis_amd.pqmf import PQMF

dev = torch.device("cuda:0")
B, T, N = int(os.environ.get("B", 128)), int(os.environ.get("T", 176400)), int(os.environ.get("N", 3))
gram = PQMF(N).to(dev)
x = torch.randn(B, 1, T, device=dev)
for _ in range(3):
    z = gram(x)
torch.cuda.synchronize()
K = 20
g = torch.cuda.CUDAGraph()
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    with torch.cuda.graph(g, stream=s):
        for _ in range(K):
            z = gram(x)
best = 1e9
for _ in range(5):
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record(); g.replay(); b.record(); torch.cuda.synchronize()
    best = min(best, a.elapsed_time(b) / K)
nbytes = 4.0 * B * T + 4.0 * z.numel()
print(f"pqmf N={N} B={B} T={T}: {best*1e3:.1f} us/launch  {nbytes/best/1e9:.2f} TB/s algorithmic")

# the same bytes as a plain copy (x -> first B*T floats of a buffer), for scale
from inverse_audio_synthesis_amd import _lib
lib = _lib.load()
dst = torch.empty(B * T, device=dev)
n = B * T - (B * T) % 4
def graph_time(fn):
    g = torch.cuda.CUDAGraph(); s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        with torch.cuda.graph(g, stream=s):
            for _ in range(K): fn()
    best = 1e9
    for _ in range(5):
        a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
        a.record(); g.replay(); b.record(); torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b) / K)
    return best
tc = graph_time(lambda: lib.ias_stream_copy(_lib.ptr(x), _lib.ptr(dst), n, _lib.stream()))
print(f"ias_stream_copy {8.0*n/1e6:.0f} MB moved: {tc*1e3:.1f} us  {8.0*n/tc/1e9:.2f} TB/s")
tt = graph_time(lambda: dst.copy_(x.view(-1)))
print(f"torch copy_: {tt*1e3:.1f} us  {8.0*n/tt/1e9:.2f} TB/s")
