"""torch.profiler view of one eager VICReg pretraining step: the aten ops behind the remaining torch launches (adds, sums,
copies), grouped by op and input shapes.  usage: python scripts/diag/profile_pretrain_ops.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from torch.profiler import profile, ProfilerActivity
from inverse_audio_synthesis_amd.config import load_config
from inverse_audio_synthesis_amd.harness import VicregAudioParams
from inverse_audio_synthesis_amd.trainer import Trainer

dev = torch.device("cuda:0")
torch.manual_seed(42)
cfg = load_config(os.path.join(ROOT, "conf"), "config", ["vicreg.batch_size=128", "trainer.cuda_graph=false"])
model = VicregAudioParams(cfg)
tr = Trainer(cfg, model, stage="vicreg", device=dev)
model.train()


def step(i):
    tr.bucketer.begin_step()
    model.training_step(i).backward()
    tr.optimizer.step()


for i in range(3):
    step(i)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    step(5)
    torch.cuda.synchronize()
rows = []
for e in prof.key_averages(group_by_input_shape=True):
    dt = getattr(e, "self_device_time_total", None)
    if dt is None:
        dt = getattr(e, "self_cuda_time_total", 0)
    if dt > 0 and e.key.startswith("aten::"):
        rows.append((dt, e.count, e.key, str(e.input_shapes)[:110]))
rows.sort(reverse=True)
for dt, cnt, key, shp in rows[:int(os.environ.get('ROWS', 45))]:
    print(f"{dt:9.1f} us  x{cnt:<3d} {key:32s} {shp}")
