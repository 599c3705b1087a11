"""Forward / backward time of every block of the AudioEmbedding trunk at batch 128 (HIP events) -- which layers are slow."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from inverse_audio_synthesis_amd.vision import mobilenet_v3_small
from inverse_audio_synthesis_amd.audioembed import conv2x2_nhwc

dev = torch.device("cuda:0")
B = int(os.environ.get("B", 128))
net = mobilenet_v3_small().to(dev).train()
x = torch.randn(B, 3, 240, 245, device=dev)
def timed(fn, n=3):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n
tot_f = tot_b = 0.0
for i, layer in enumerate(net.features):
    xin = x.detach().clone().requires_grad_(True)
    y = layer(xin)
    g = torch.randn_like(y)
    tf = timed(lambda: layer(xin))
    def fb():
        yy = layer(xin)
        torch.autograd.grad(yy, [xin] + list(layer.parameters()), g)
    tb = timed(fb) - tf
    tot_f += tf; tot_b += tb
    desc = type(layer).__name__
    convs = [(m.in_channels, m.out_channels, m.kernel_size[0], m.stride[0], m.groups) for m in layer.modules() if isinstance(m, torch.nn.Conv2d)]
    print(f"features.{i:2d} {desc:18s} in {tuple(xin.shape)} -> {tuple(y.shape)}  fwd {tf:7.3f} ms  bwd {tb:7.3f} ms  convs {convs}")
    x = y.detach()
print(f"MobileNet body total: fwd {tot_f:.2f} ms, bwd {tot_b:.2f} ms")
