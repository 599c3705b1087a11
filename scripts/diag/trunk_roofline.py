"""Every leaf op of the MobileNetV3 trunk at batch 128 (depthwise / thin 1x1 / stem convolutions, BatchNorm + activation,
squeeze-excitation), timed alone, forward and backward, against the bytes it has to move: which layers sit far above
their HBM time.  Developer tool (one GPU):  python scripts/diag/trunk_roofline.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from inverse_audio_synthesis_amd import vision
from inverse_audio_synthesis_amd.vision import mobilenet_v3_small

dev = torch.device("cuda:0")
B = int(os.environ.get("B", 128))
net = mobilenet_v3_small().to(dev).train()
LEAVES = (vision.DepthwiseConv2d, vision.PointwiseConv2d, vision.StemConv2d, vision.BatchNormAct2d, vision.SqueezeExcitation)
seen = []
def hook(name):
    def f(mod, inp, out):
        seen.append((name, mod, tuple(inp[0].shape), tuple(out.shape)))
    return f
hs = [m.register_forward_hook(hook(n)) for n, m in net.named_modules() if isinstance(m, LEAVES) and not any(
    isinstance(p, vision.SqueezeExcitation) and p is not m for pn, p in net.named_modules() if n.startswith(pn + "."))]
with torch.no_grad():
    net(torch.randn(B, 3, 240, 245, device=dev)) if not hasattr(net, "features") else net.features(torch.randn(B, 3, 240, 245, device=dev))
for h in hs: h.remove()

def timed(fn, n=10):
    """us per call: n calls captured into one hipGraph (no host launch cost in the number), fastest of 5 replays"""
    for _ in range(2): fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n): fn()
    g.replay(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(5):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); g.replay(); b.record(); torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b) / n)
    return best * 1e3   # us

rows = []
for name, mod, si, so in seen:
    x = torch.randn(si, device=dev, requires_grad=True)
    y = mod(x)
    g = torch.randn_like(y)
    params = [p for p in mod.parameters()]
    tf = timed(lambda: mod(x))
    def fb():
        torch.autograd.grad(mod(x), [x] + params, g)
    tb = max(timed(fb) - tf, 0.0)
    nin, nout = x.numel() * 4, y.numel() * 4
    kind = type(mod).__name__
    if kind == "BatchNormAct2d":
        bf, bb = 2 * nin + nout, 4 * nin + nin          # stats pass + apply; sums pass (x, g) + apply (x, g -> dx)
    elif kind == "SqueezeExcitation":
        bf, bb = 2 * nin + nout, 4 * nin + nin          # pool + scale; <g, x> pass + scale pass (g, -> gx)
    else:
        bf, bb = nin + nout, (nout + nin) + (nin + nout)   # data gradient (g -> gx) + weight gradient (x, g)
    rows.append((name, kind, si, so, tf, bf, tb, bb))
HBM = 5.0e6   # bytes per us of a well-formed streaming kernel on this part (~5 TB/s)
print(f"{'layer':34s} {'kind':18s} {'in':22s} fwd us (ideal)  GB/s | bwd us (ideal)  GB/s | excess us")
tot = [0.0, 0.0, 0.0, 0.0]
by_kind = {}
for name, kind, si, so, tf, bf, tb, bb in rows:
    ex = (tf - bf / HBM) + (tb - bb / HBM)
    tot[0] += tf; tot[1] += bf / HBM; tot[2] += tb; tot[3] += bb / HBM
    k = by_kind.setdefault(kind, [0.0, 0.0])
    k[0] += tf + tb; k[1] += (bf + bb) / HBM
    print(f"{name:34s} {kind:18s} {str(si):22s} {tf:7.1f} ({bf / HBM:6.1f}) {bf / tf / 1e3:6.0f} | {tb:7.1f} ({bb / HBM:6.1f}) {bb / max(tb, 1e-3) / 1e3:6.0f} | {ex:7.1f}")
print(f"total: fwd {tot[0]:.0f} us (ideal {tot[1]:.0f}), bwd {tot[2]:.0f} us (ideal {tot[3]:.0f})")
for k, v in by_kind.items():
    print(f"  {k:18s} {v[0]:7.0f} us, ideal {v[1]:6.0f} us, excess {v[0] - v[1]:6.0f}")
