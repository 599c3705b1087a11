"""STFT kernel at the headline shape (B=128 x 176400, n_fft 1024, hop 512): mel-L1 loss mode, mel output mode, raw bins."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from inverse_audio_synthesis_amd.spectral import MelSpectrogramL1, STFTPlan, VALUE_POWER

dev = torch.device("cuda:0")
B = 128
a = (torch.randn(B, 176400, generator=torch.Generator().manual_seed(0)) * 0.1).to(dev)
mel = MelSpectrogramL1().to(dev)
tm = mel.target(a).clone()
raw = STFTPlan(1024, None, 512).to(dev)
def timed(name, fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n): fn()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
    print(f"{name:40s} {e0.elapsed_time(e1) / n * 1e3:8.1f} us")
parts = os.environ.get("PARTS", "loss,mel,raw").split(",")      # PARTS=raw: one variant only (for per-variant PMC runs)
if "loss" in parts:
    timed("mel-L1 vs cached target (loss mode)", lambda: mel(a, target_mel=tm))
if "mel" in parts:
    timed("mel spectrogram out (128 mels)", lambda: mel.mel.plan.values(a, VALUE_POWER))
if "raw" in parts:
    timed("power spectrogram out (513 bins)", lambda: raw.values(a, VALUE_POWER))
