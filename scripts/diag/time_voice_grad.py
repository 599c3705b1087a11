"""Audio-rate Voice backward alone at the configs[4] shape (B=64 x 4 s @ 44.1 kHz): time per call and a checksum of its
outputs.  IAS_VOICE_GRAD_V1=1 selects the chunk-scan kernels (A/B).  usage: python scripts/diag/time_voice_grad.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from inverse_audio_synthesis_amd.voice import SynthConfig, Voice
from inverse_audio_synthesis_amd.voice_grad import audio_rate_backward, normalisation_rows

dev = torch.device("cuda:0")
B = int(os.environ.get("B", 64))
v = Voice(SynthConfig(batch_size=B, reproducible=False)).to(dev)
v.randomize(3)
audio = v.render()
g = torch.randn(audio.shape, generator=torch.Generator().manual_seed(1)).to(dev)
rn = normalisation_rows(g, audio, v.peaks_view()) if hasattr(v, "peaks_view") else None
ctl = v.rendered_control()
for _ in range(3):
    gc, gs = audio_rate_backward(v, v.params01, g, rn, ctl)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
N = 20
e0.record()
for _ in range(N):
    gc, gs = audio_rate_backward(v, v.params01, g, rn, ctl)
e1.record()
torch.cuda.synchronize()
print(f"voice backward B={B}: {e0.elapsed_time(e1) / N * 1e3:.1f} us  g_ctrl {gc.double().abs().sum().item():.9e}  "
      f"g_scal {gs.abs().sum().item():.9e}  finite {bool(torch.isfinite(gc).all() and torch.isfinite(gs).all())}")
