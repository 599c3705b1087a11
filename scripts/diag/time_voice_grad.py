"""Voice render forward + backward at the headline size (HIP events) and gradient error vs the fp64 oracle on a small case."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from inverse_audio_synthesis_amd.voice import SynthConfig, Voice
from oracle import synth_oracle as so

dev = torch.device("cuda:0")
for (B, sr, sec, seed) in ((4, 16000, 1.0, 0), (4, 16000, 1.0, 2), (2, 44100, 4.0, 3)):
    cfg = so.VoiceConfig(B, sr, sec)
    v = Voice(SynthConfig(batch_size=B, sample_rate=sr, buffer_size_seconds=sec, reproducible=False)).to(dev)
    p0 = so.sample_params01(cfg, seed)
    w = torch.randn((B, cfg.buffer_size), generator=torch.Generator().manual_seed(100 + seed))
    pd = p0.double().requires_grad_(True)
    (ref,) = torch.autograd.grad((so.render_from_params01(cfg, pd, so.make_noise(cfg), "f64") * w.double()).sum(), pd)
    p = p0.to(dev).requires_grad_(True)
    (v.render(p) * w.to(dev)).sum().backward()
    g = p.grad.cpu().double()
    rel = [((g[b] - ref[b]).norm() / ref[b].norm()).item() for b in range(B)]
    print(f"B={B} sr={sr} sec={sec} seed={seed}: rel L2 per voice {['%.1e' % r for r in rel]} batch {((g - ref).norm() / ref.norm()).item():.1e}")

B = int(os.environ.get("B", 128))
v = Voice(SynthConfig(batch_size=B, reproducible=False)).to(dev)
p = torch.rand(B, 78, generator=torch.Generator().manual_seed(1000)).to(dev)
for _ in range(2):
    q = p.clone().requires_grad_(True); a = v.render(q); a.square().mean().backward()
torch.cuda.synchronize()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
q = p.clone().requires_grad_(True)
ev[0].record(); a = v.render(q); ev[1].record(); loss = a.square().mean(); ev[2].record(); loss.backward(); ev[3].record()
torch.cuda.synchronize()
print(f"B={B} 4 s @ 44.1 kHz: forward {ev[0].elapsed_time(ev[1]):.3f} ms, loss {ev[1].elapsed_time(ev[2]):.3f} ms, backward {ev[2].elapsed_time(ev[3]):.3f} ms")
