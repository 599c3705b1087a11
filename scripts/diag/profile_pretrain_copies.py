"""Which call sites issue the copy / elementwise kernels of one pretraining step (torch.profiler with stacks) -- developer tool."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from torch.profiler import profile, ProfilerActivity
from inverse_audio_synthesis_amd.config import load_config
from inverse_audio_synthesis_amd.harness import VicregAudioParams

dev = torch.device("cuda:0")
cfg = load_config(os.path.join(ROOT, "conf"), "config", ["vicreg.batch_size=128", "trainer.cuda_graph=false"])
model = VicregAudioParams(cfg).to(dev).train()
opt = model.configure_optimizers()
opt = opt["optimizer"] if isinstance(opt, dict) else opt
def step(i):
    loss = model.training_step(i)
    opt.zero_grad(set_to_none=True)
    loss.backward()
    opt.step()
for i in range(3): step(i)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
    step(10)
    torch.cuda.synchronize()
print(prof.key_averages(group_by_input_shape=True).table(sort_by="cuda_time_total", row_limit=45, max_name_column_width=60))
print(prof.key_averages(group_by_stack_n=6).table(sort_by="cuda_time_total", row_limit=30, max_name_column_width=50))
