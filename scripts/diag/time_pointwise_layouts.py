"""Times the trunk's 1x1 convolutions (MobileNetV3-small at [128,3,240,245]) in two activation layouts: NCHW as a 128-batch
bmm with the weight expanded (what vision.PointwiseConv2d does) and channel-major [C, B*HW] as one mm; forward + both
gradients through autograd; third column: vision.PointwiseConv2d as shipped (csrc/pointwise_kernels.hip where supported).  usage: python scripts/diag/time_pointwise_layouts.py"""
import torch

B = 128
# (Cin, Cout, H, W) of every 1x1 convolution of the trunk (expand / project of each block, the last 96 -> 576)
LAYERS = [(16, 16, 60, 62), (16, 72, 60, 62), (72, 24, 30, 31), (24, 88, 30, 31), (88, 24, 30, 31), (24, 96, 30, 31),
          (96, 40, 15, 16), (40, 240, 15, 16), (240, 40, 15, 16), (40, 240, 15, 16), (240, 40, 15, 16), (40, 120, 15, 16),
          (120, 48, 15, 16), (48, 144, 15, 16), (144, 48, 15, 16), (48, 288, 15, 16), (288, 96, 8, 8), (96, 576, 8, 8),
          (576, 96, 8, 8), (96, 576, 8, 8), (576, 96, 8, 8), (96, 576, 8, 8)]


def timed(fn, n=20):
    """GPU time of fn per call: fn is captured into a hipGraph (its host cost, ~150 us of autograd, is not the subject)."""
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3):
            fn()
    torch.cuda.current_stream().wait_stream(side)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        fn()
    graph.replay()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        graph.replay()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


def main():
    dev = torch.device("cuda:0")
    import os, sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
    import importlib
    PointwiseConv2d = importlib.import_module('inverse_audio_synthesis_amd.vision').PointwiseConv2d
    tot = [0.0, 0.0, 0.0]
    for (ci, co, h, w) in LAYERS:
        hw = h * w
        wt = torch.randn(co, ci, device=dev, requires_grad=True)
        x1 = torch.randn(B, ci, hw, device=dev, requires_grad=True)
        g1 = torch.randn(B, co, hw, device=dev)
        x2 = torch.randn(ci, B * hw, device=dev, requires_grad=True)
        g2 = torch.randn(co, B * hw, device=dev)

        def f1():
            y = torch.bmm(wt.view(1, co, ci).expand(B, -1, -1), x1)
            torch.autograd.grad(y, (x1, wt), g1)

        def f2():
            y = torch.mm(wt, x2)
            torch.autograd.grad(y, (x2, wt), g2)

        conv = PointwiseConv2d(ci, co, 1, bias=False).to(dev)
        x3 = torch.randn(B, ci, h, w, device=dev, requires_grad=True)
        g3 = g1.view(B, co, h, w)

        def f3():
            torch.autograd.grad(conv(x3), (x3, conv.weight), g3)

        t1, t2, t3 = timed(f1), timed(f2), timed(f3)
        tot[0] += t1; tot[1] += t2; tot[2] += t3
        print(f"{ci:4d} -> {co:4d}  HW {hw:5d}:  bmm {t1:8.1f} us   mm {t2:8.1f} us   PointwiseConv2d {t3:8.1f} us", flush=True)
    print(f"total: bmm {tot[0] / 1e3:.2f} ms, mm {tot[1] / 1e3:.2f} ms, PointwiseConv2d {tot[2] / 1e3:.2f} ms "
          f"(forward + input gradient + weight gradient, B={B})")


if __name__ == "__main__":
    main()
