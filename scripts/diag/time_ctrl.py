"""The control pass alone: K launches in a hipGraph (developer tool; IAS_HIP_LIB=...libias_hip_diag.so + IAS_VOICE_CTRL=libm|fused select the other forms)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from inverse_audio_synthesis_amd.voice import SynthConfig, Voice
dev = torch.device("cuda:0")
B, K = int(os.environ.get("B", 128)), 20
voice = Voice(SynthConfig(batch_size=B, reproducible=False)).to(dev)
voice.set_parameters01(torch.rand(B, 78, generator=torch.Generator().manual_seed(1000)).to(dev))
ws = voice.new_workspace(dev)
voice.render_control(ws); torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    for _ in range(K): voice.render_control(ws)
g.replay(); torch.cuda.synchronize()
ts = []
for _ in range(9):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1) / K * 1e3)
ts.sort()
print(f"control pass B={B}: median {ts[4]:.1f} us  min {ts[0]:.1f} us  lib={os.environ.get('IAS_HIP_LIB', 'product')} form={os.environ.get('IAS_VOICE_CTRL', 'slim')}")
