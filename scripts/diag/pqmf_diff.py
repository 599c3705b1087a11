"""Where the MFMA analysis differs from the VALU kernels (misaligned view) -- developer tool."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from inverse_audio_synthesis_amd import _lib
from inverse_audio_synthesis_amd.pqmf import PQMF
lib = _lib.load()
dev = torch.device("cuda:0")
N, B, T = int(os.environ.get("N", 3)), int(os.environ.get("B", 5)), int(os.environ.get("T", 176400))
m = PQMF(N).to(dev)
Hc = m.H.reshape(N, 63).contiguous()
packed = torch.empty(lib.ias_pqmf_packed_taps_len(N, 63), device=dev)
lib.ias_pqmf_pack_taps(_lib.ptr(Hc), _lib.ptr(packed), N, 63, _lib.stream())
torch.manual_seed(0)
x = torch.randn(B, T, device=dev)
xm = torch.empty(B * T + 1, device=dev)[1:]
xm.copy_(x.flatten())
L = lib.ias_pqmf_out_len(T, N, 63)
mean = torch.linspace(-0.1, 0.1, N, device=dev); std = torch.linspace(0.5, 1.5, N, device=dev)
peak = torch.linspace(0.5, 3.0, B, device=dev)
for fused in (False, True):
    za = torch.full((B, N, L), 7.0, device=dev); zm = torch.full((B, N, L), 9.0, device=dev)
    args = (_lib.ptr(mean), _lib.ptr(std), _lib.ptr(peak)) if fused else (None, None, None)
    lib.ias_pqmf_analysis(_lib.ptr(x), _lib.ptr(Hc), _lib.ptr(packed), None, _lib.ptr(za), *args, B, T, N, 63, _lib.stream())
    lib.ias_pqmf_analysis(_lib.ptr(xm), _lib.ptr(Hc), _lib.ptr(packed), None, _lib.ptr(zm), *args, B, T, N, 63, _lib.stream())
    torch.cuda.synchronize()
    d = (za != zm)
    print("fused", fused, "ndiff", int(d.sum()), "of", d.numel(), "max", float((za - zm).abs().max()))
    if d.any():
        idx = d.nonzero()
        print(idx[:10].tolist(), idx[-5:].tolist())
        for i in idx[:5]:
            print(tuple(i.tolist()), float(za[tuple(i)]), float(zm[tuple(i)]))
        fr = idx[:, 2]
        print("frames mod 320 hist:", torch.bincount(fr % 320, minlength=320).nonzero().flatten()[:40].tolist())
        print("rows:", torch.bincount(idx[:, 0], minlength=B).tolist(), "bands:", torch.bincount(idx[:, 1], minlength=N).tolist())
