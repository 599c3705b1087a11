"""One VICReg pretraining step (BASELINE config #3: B=128, 4 s @ 44.1 kHz, dim 1024, embeddim 8192) -- wall time per step,
eager (default) or as the Trainer's captured hipGraph (GRAPH=1); DEFER=0: without the joint weight-gradient reduction;
SEPROJ=0: without the squeeze-excitation gate taken by the projection on load; SEPOOL=0: without the gate's pool taken by the
normalisation in front of it."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from inverse_audio_synthesis_amd.config import load_config
from inverse_audio_synthesis_amd.harness import VicregAudioParams
from inverse_audio_synthesis_amd.trainer import Trainer

dev = torch.device("cuda:0")
B = int(os.environ.get("B", 128))
graph = bool(int(os.environ.get("GRAPH", "0")))
torch.manual_seed(42)
cfg = load_config(os.path.join(ROOT, "conf"), "config", [f"vicreg.batch_size={B}", f"trainer.cuda_graph={'true' if graph else 'false'}"])
model = VicregAudioParams(cfg)
tr = Trainer(cfg, model, stage="vicreg", device=dev)
if os.environ.get("DEFER") == "0":      # the trunk's weight-gradient reductions as a launch per layer (vision.defer_weight_reductions off)
    import contextlib
    tr._deferred_reductions = contextlib.nullcontext
if os.environ.get("SEPROJ") == "0":     # SqueezeExcitation and its projection as two nodes (the gate's own pass over the map)
    from inverse_audio_synthesis_amd import vision
    vision.FUSE_SE_PROJECTION = False
if os.environ.get("SEPOOL") == "0":     # the squeeze-excitation gate pools its input itself (no pool from the normalisation in front)
    from inverse_audio_synthesis_amd import vision
    vision.SE_POOL_FROM_NORM = False
model.train()
opt = tr.optimizer
def step(i):
    if graph:
        tr._graph_step(i, i)
    else:
        tr.bucketer.begin_step()
        model.training_step(i).backward(tr._seed())
        opt.step()
for i in range(5): step(i)          # (graph mode: 3 eager warm-up steps, capture, first replays)
torch.cuda.synchronize()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
n = int(os.environ.get("STEPS", 5))
ev[0].record()
for i in range(n): step(10 + i)
ev[1].record(); torch.cuda.synchronize()
loss = model.logged["vicreg/train/loss"]
print(f"VICReg pretraining step B={B} ({'hipGraph replay' if graph else 'eager'}): {ev[0].elapsed_time(ev[1]) / n:.2f} ms/step "
      f"(render + PQMF + MobileNetV3 trunk + projector + loss + backward + LARS), loss {loss.item():.4f}")
