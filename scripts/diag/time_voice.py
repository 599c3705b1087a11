"""Isolated timing of the audio-rate render stage (HIP events around K back-to-back launches in one hipGraph).
Env: B (128), K (20), MATH (0|1), IAS_HIP_LIB (alternative build of the same C ABI), EAGER=1 (no graph: for rocprofv3)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from inverse_audio_synthesis_amd.voice import SynthConfig, Voice

dev = torch.device("cuda:0")
B, K = int(os.environ.get("B", 128)), int(os.environ.get("K", 20))
voice = Voice(SynthConfig(batch_size=B, reproducible=False)).to(dev)
voice.math_mode = int(os.environ.get("MATH", 0))
voice.set_parameters01(torch.rand(B, 78, generator=torch.Generator().manual_seed(1000)).to(dev))
ws = voice.new_workspace(dev)
audio = torch.empty((B, voice.synthconfig.buffer_size), dtype=torch.float32, device=dev)
voice.render_control(ws)


def run(k):
    for _ in range(k):
        voice.render_audio(ws, out=audio, normalize=False)


run(3)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
if os.environ.get("EAGER"):
    e0.record(); run(K); e1.record()
else:
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        run(K)
    g.replay(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(5):
        e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / K)
torch.cuda.synchronize()
us = (e0.elapsed_time(e1) / K if os.environ.get("EAGER") else best) * 1e3
print(f"voice audio stage B={B} math={voice.math_mode} lib={os.environ.get('IAS_HIP_LIB', 'default')}: {us:.1f} us/launch "
      f"status={voice.chain_status(ws)} checksum={audio.double().abs().sum().item():.6f}")
