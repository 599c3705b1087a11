"""Timing of the VICReg loss kernels (developer tool)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from inverse_audio_synthesis_amd.vicreg import vicreg_loss
dev = torch.device("cuda:0")
for Bv in (128, 1024):
    x = torch.randn(Bv, 8192, device=dev); y = torch.randn(Bv, 8192, device=dev)
    for _ in range(3): vicreg_loss(x, y, Bv)
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(20): out = vicreg_loss(x, y, Bv)
    e.record(); torch.cuda.synchronize()
    print(f"vicreg_loss B={Bv}: {s.elapsed_time(e)/20*1e3:.1f} us  cov={out[3].item():.6f}")
    # accuracy of the bf16 Gram vs fp64 on the same data
    xc = (x - x.mean(0)).double(); yc = (y - y.mean(0)).double()
    def cov_loss(v):
        c = (v.T @ v) / (Bv - 1); return (c.pow(2).sum() - c.diagonal().pow(2).sum()) / 8192
    ref = (cov_loss(xc) + cov_loss(yc)).item()
    print(f"   cov_loss rel err vs fp64: {abs(out[3].item()-ref)/ref:.2e}")
