"""Per-stage timing of the bench step (HIP events, eager launches) -- developer tool."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from inverse_audio_synthesis_amd.voice import SynthConfig, Voice
from inverse_audio_synthesis_amd.pqmf import PQMF
from inverse_audio_synthesis_amd.spectral import MelSpectrogramL1
from inverse_audio_synthesis_amd.vicreg import vicreg_loss

dev = torch.device("cuda:0")
B = int(os.environ.get("B", 128))
cfg = SynthConfig(batch_size=B, reproducible=False)
voice = Voice(cfg).to(dev); gram = PQMF(3).to(dev); mel = MelSpectrogramL1().to(dev)
params = torch.rand(B, 78, generator=torch.Generator().manual_seed(1000)).to(dev)
voice.set_parameters01(params)
tm = mel.target(voice.render(torch.rand(B, 78, generator=torch.Generator().manual_seed(2000)).to(dev))).clone()
times = {}
def hook(name, phase):
    e = torch.cuda.Event(enable_timing=True); e.record(); times.setdefault(name, {}).setdefault(phase, []).append(e)
def stage(name, fn):
    hook(name, "begin"); r = fn(); hook(name, "end"); return r
iters = int(os.environ.get("ITERS", 30))
for it in range(iters + 5):
    if it == 5:
        times.clear()
    audio = voice.render_staged(on_stage=hook)
    z = stage("pqmf", lambda: gram(audio.unsqueeze(1)))
    loss = stage("mel_l1", lambda: mel(audio, target_mel=tm))
torch.cuda.synchronize()
tot = 0.0
for name, d in times.items():
    ms = [b.elapsed_time(e) for b, e in zip(d["begin"], d["end"])]
    avg = sum(ms) / len(ms); tot += avg
    print(f"{name:12s} {avg*1e3:9.1f} us  (min {min(ms)*1e3:.1f})")
print(f"{'sum':12s} {tot*1e3:9.1f} us   loss={loss.item():.6f}")
if os.environ.get("VICREG"):
    for Bv in (128, 1024):
        x = torch.randn(Bv, 8192, device=dev); y = torch.randn(Bv, 8192, device=dev)
        for _ in range(3): vicreg_loss(x, y, Bv)
        s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(10): out = vicreg_loss(x, y, Bv)
        e.record(); torch.cuda.synchronize()
        print(f"vicreg_loss B={Bv}: {s.elapsed_time(e)/10*1e3:.1f} us  cov={out[3].item():.6f}")
