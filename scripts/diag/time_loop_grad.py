"""params -> Voice render -> mel-L1 -> backward to params at the headline size: stage times (HIP events)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from inverse_audio_synthesis_amd.voice import SynthConfig, Voice
from inverse_audio_synthesis_amd.spectral import MelSpectrogramL1

dev = torch.device("cuda:0")
B = int(os.environ.get("B", 128))
v = Voice(SynthConfig(batch_size=B, reproducible=False)).to(dev)
mel = MelSpectrogramL1().to(dev)
p = torch.rand(B, 78, generator=torch.Generator().manual_seed(1000)).to(dev)
tm = mel.target(v.render(torch.rand(B, 78, generator=torch.Generator().manual_seed(2000)).to(dev))).clone()
for it in range(4):
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
    q = p.clone().requires_grad_(True)
    ev[0].record(); a = v.render(q); ev[1].record(); loss = mel(a, target_mel=tm); ev[2].record(); loss.backward(); ev[3].record()
    torch.cuda.synchronize()
print(f"B={B}: render {ev[0].elapsed_time(ev[1]):.3f} ms, mel-L1 {ev[1].elapsed_time(ev[2]):.3f} ms, backward (mel + synth + control graph) {ev[2].elapsed_time(ev[3]):.3f} ms; loss {loss.item():.5f} |g| {q.grad.norm().item():.4e}")
