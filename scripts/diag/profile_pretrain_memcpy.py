"""Which Python call sites issue the device-to-device copies (aten::copy_ / clone -> __amd_rocclr_copyBuffer), fills and
small torch reductions of one pretraining step: torch.profiler with stacks, aggregated by (op, innermost package frame).
Developer tool."""
import os, sys, collections
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from torch.profiler import profile, ProfilerActivity
from inverse_audio_synthesis_amd.config import load_config
from inverse_audio_synthesis_amd.harness import VicregAudioParams
from inverse_audio_synthesis_amd.trainer import Trainer

dev = torch.device("cuda:0")
cfg = load_config(os.path.join(ROOT, "conf"), "config", ["vicreg.batch_size=128", "trainer.cuda_graph=false"])
model = VicregAudioParams(cfg)
tr = Trainer(cfg, model, stage="vicreg", device=dev)
model.train()
def step(i):
    tr.bucketer.begin_step()
    model.training_step(i).backward()
    tr.optimizer.step()
for i in range(3): step(i)
torch.cuda.synchronize()
import traceback
from torch.utils._python_dispatch import TorchDispatchMode
agg = collections.Counter()
WANT = ("copy_", "clone", "fill_", "zero_", "sum", "add", "add_", "mul", "cat", "contiguous", "mean", "div", "div_", "mul_", "_to_copy",
        "sub", "neg", "ones_like", "zeros_like", "zeros", "sqrt", "where", "index_select", "native_batch_norm", "relu", "threshold_backward")
class Log(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = func.__name__.split(".")[0]
        if name in WANT:
            st = [f for f in traceback.extract_stack() if "audio-synthesis_amd/" in f.filename or "audio_synthesis_amd/" in f.filename]
            site = f"{os.path.basename(st[-1].filename)}:{st[-1].lineno} {st[-1].name}" if st else "(no package frame: autograd engine / torch module)"
            shp = tuple(args[0].shape) if args and hasattr(args[0], "shape") else ()
            agg[(name, site, shp if len(shp) < 3 else shp[:1] + ("...",))] += 1
        return func(*args, **(kwargs or {}))
with Log():
    step(10)
torch.cuda.synchronize()
for (name, site, shp), n in sorted(agg.items(), key=lambda kv: -kv[1])[:90]:
    print(f"{n:4d}  {name:18s} {str(shp):22s} {site}")
