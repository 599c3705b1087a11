"""Does ias_vicreg_loss read workspace bytes it has not written?  The same call on a workspace pre-filled with 0x00, 0xFF
(NaN patterns) and 0x3F bytes.  usage (GPU box): python scripts/diag/dbg_vicreg_ws.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from inverse_audio_synthesis_amd import _lib
lib = _lib.load()
dev = torch.device("cuda:0")
for B, D in ((128, 8192), (1024, 8192), (64, 512), (96, 1024), (8, 64)):
    x = torch.randn(B, D, generator=torch.Generator().manual_seed(0)).to(dev)
    y = torch.randn(B, D, generator=torch.Generator().manual_seed(1)).to(dev)
    need = int(lib.ias_vicreg_workspace_bytes(B, D))
    for fill in (0x00, 0xFF, 0x3F):
        ws = torch.full((need + 4096,), fill, dtype=torch.uint8, device=dev)
        out = torch.full((4,), float("nan"), device=dev)
        st = lib.ias_vicreg_loss(_lib.ptr(x), _lib.ptr(y), _lib.ptr(out), _lib.ptr(ws), need, B, D, B, 25.0, 25.0, 1.0, _lib.stream())
        gx, gy = torch.empty_like(x), torch.empty_like(y)
        one = torch.ones((), device=dev)
        st2 = lib.ias_vicreg_backward4_ld(_lib.ptr(x), _lib.ptr(y), D, one.data_ptr(), None, None, None, _lib.ptr(gx), _lib.ptr(gy), D,
                                          _lib.ptr(ws), need, B, D, B, 25.0, 25.0, 1.0, _lib.stream())
        torch.cuda.synchronize()
        tail_ok = bool((ws[need:] == fill).all())
        print(f"B={B} D={D} fill={fill:#04x}: st {st} {st2} out {[round(v, 6) for v in out.tolist()]} |gx| {float(gx.double().norm()):.6f} "
              f"|gy| {float(gy.double().norm()):.6f} bytes past the workspace untouched: {tail_ok}", flush=True)
