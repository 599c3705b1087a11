"""mel-L1 kernel alone (hipGraph of 20 launches, B=128 x 4 s) -- developer tool."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from inverse_audio_synthesis_amd.spectral import MelSpectrogramL1
dev = torch.device("cuda:0")
B, T = int(os.environ.get("B", 128)), 176400
mel = MelSpectrogramL1().to(dev)
x = torch.randn(B, T, device=dev) * 0.3
tm = mel.target(torch.randn(B, T, device=dev) * 0.3).clone()
for _ in range(3): l = mel(x, target_mel=tm)
torch.cuda.synchronize()
K = 20
g = torch.cuda.CUDAGraph(); s = torch.cuda.Stream()
with torch.cuda.stream(s):
    with torch.cuda.graph(g, stream=s):
        for _ in range(K): l = mel(x, target_mel=tm)
best = 1e9
for _ in range(5):
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record(); g.replay(); b.record(); torch.cuda.synchronize()
    best = min(best, a.elapsed_time(b) / K)
print(f"mel-L1 B={B}: {best*1e3:.1f} us/launch (stft + reduce)  loss {l.item():.6f}")
