"""One VICReg pretraining step (BASELINE config #3: B=128, 4 s @ 44.1 kHz, dim 1024, embeddim 8192) -- stage times."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from inverse_audio_synthesis_amd.config import load_config
from inverse_audio_synthesis_amd.harness import VicregAudioParams

dev = torch.device("cuda:0")
B = int(os.environ.get("B", 128))
cfg = load_config(os.path.join(ROOT, "conf"), "config", [f"vicreg.batch_size={B}"])
model = VicregAudioParams(cfg).to(dev).train()
opt = model.configure_optimizers()
opt = opt["optimizer"] if isinstance(opt, dict) else opt
def step(i):
    loss = model.training_step(i)
    opt.zero_grad(set_to_none=True)
    loss.backward()
    opt.step()
    return loss
for i in range(3): step(i)
torch.cuda.synchronize()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
ev[0].record()
for i in range(5): loss = step(10 + i)
ev[1].record(); torch.cuda.synchronize()
print(f"VICReg pretraining step B={B}: {ev[0].elapsed_time(ev[1]) / 5:.2f} ms/step (render + PQMF + MobileNetV3 trunk + projector + loss + backward + LARS), loss {loss.item():.4f}")
