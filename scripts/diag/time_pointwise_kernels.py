"""Per-call times of csrc/pointwise_kernels.hip (forward, input gradient, weight gradient) for the trunk's 1x1 convolutions
at B=128, through the C ABI.  usage: python scripts/diag/time_pointwise_kernels.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from inverse_audio_synthesis_amd import _lib

B = 128
LAYERS = [(16, 16, 3720), (16, 72, 3720), (72, 24, 930), (24, 88, 930), (88, 24, 930), (24, 96, 930), (96, 40, 240),
          (40, 240, 240), (240, 40, 240), (40, 120, 240), (120, 48, 240), (48, 144, 240), (144, 48, 240), (48, 288, 240)]


def timed(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


def main():
    lib = _lib.load()
    dev = torch.device("cuda:0")
    st = _lib.stream()
    tot = [0.0, 0.0, 0.0]
    for ci, co, hw in LAYERS:
        x = torch.randn(B, ci, hw, device=dev)
        g = torch.randn(B, co, hw, device=dev)
        w = torch.randn(co, ci, device=dev)
        y, gx, gw = torch.empty_like(g), torch.empty_like(x), torch.empty_like(w)
        scratch = torch.empty(int(lib.ias_pwconv_weight_scratch(B, ci, co, hw)), device=dev)
        t = [timed(lambda: lib.ias_pwconv_forward(_lib.ptr(x), _lib.ptr(w), _lib.ptr(y), B, ci, co, hw, st)),
             timed(lambda: lib.ias_pwconv_backward_data(_lib.ptr(g), _lib.ptr(w), _lib.ptr(gx), B, ci, co, hw, st)),
             timed(lambda: lib.ias_pwconv_backward_weight(_lib.ptr(g), _lib.ptr(x), _lib.ptr(gw), _lib.ptr(scratch), B, ci, co, hw, st))]
        mb = (x.numel() + g.numel()) * 4 / 1e6
        for i in range(3):
            tot[i] += t[i]
        print(f"{ci:4d} -> {co:4d} HW {hw:5d}: forward {t[0]:7.1f} us  input grad {t[1]:7.1f} us  weight grad {t[2]:7.1f} us   "
              f"(x + y = {mb:6.1f} MB = {mb / 8e3 * 1e3:5.1f} us at 8 TB/s)", flush=True)
    print(f"total: forward {tot[0] / 1e3:.3f} ms, input grad {tot[1] / 1e3:.3f} ms, weight grad {tot[2] / 1e3:.3f} ms")


if __name__ == "__main__":
    main()
