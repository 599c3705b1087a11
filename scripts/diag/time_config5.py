"""BASELINE config #5 pieces on one GPU's share (B=64): render, PQMF(64) analysis + synthesis, 3-resolution STFT loss."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from inverse_audio_synthesis_amd.voice import SynthConfig, Voice
from inverse_audio_synthesis_amd.pqmf import PQMF
from inverse_audio_synthesis_amd.spectral import MultiResolutionSTFTLoss

dev = torch.device("cuda:0")
B = 64
v = Voice(SynthConfig(batch_size=B, reproducible=False)).to(dev)
gram = PQMF(64).to(dev)
mr = MultiResolutionSTFTLoss().to(dev)
p = torch.rand(B, 78, generator=torch.Generator().manual_seed(1)).to(dev)
tgt = v.render(torch.rand(B, 78, generator=torch.Generator().manual_seed(2)).to(dev)).clone()
def timed(name, fn, n=5):
    for _ in range(2): out = fn()
    torch.cuda.synchronize()
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): out = fn()
    b.record(); torch.cuda.synchronize()
    print(f"{name:24s} {a.elapsed_time(b) / n * 1e3:9.1f} us")
    return out
audio = timed("render", lambda: v.render(p))
z = timed("pqmf64 analysis", lambda: gram(audio.unsqueeze(1)))
timed("pqmf64 synthesis", lambda: gram.synthesis(z))
timed("mrstft loss (3 res)", lambda: mr(audio, tgt))
tg = mr.target(tgt)
timed("mrstft, cached target", lambda: mr(audio, targets=tg))
