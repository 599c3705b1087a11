"""What the headline step's kernels cost alone, side by side and one after the other (hipGraphs of K launches) --
developer tool.  Answers: do PQMF and STFT overlap when issued on two streams (the step does that), and what is the
sum the step is compared with."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from inverse_audio_synthesis_amd.voice import SynthConfig, Voice
from inverse_audio_synthesis_amd.pqmf import PQMF
from inverse_audio_synthesis_amd.spectral import MelSpectrogramL1

dev = torch.device("cuda:0")
B, K = int(os.environ.get("B", 128)), int(os.environ.get("K", 20))
cfg = SynthConfig(batch_size=B, reproducible=False)
voice = Voice(cfg).to(dev); gram = PQMF(3).to(dev); mel = MelSpectrogramL1().to(dev)
voice.set_parameters01(torch.rand(B, 78, generator=torch.Generator().manual_seed(1000)).to(dev))
tm = mel.target(voice.render(torch.rand(B, 78, generator=torch.Generator().manual_seed(2000)).to(dev))).clone()
ws = voice.new_workspace(dev)
audio = torch.empty((B, cfg.buffer_size), device=dev)
voice.render_control(ws)
voice.render_audio(ws, out=audio, normalize=False)
peaks = voice.peaks_view(ws)
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()

def pq(): return gram.analysis(audio.unsqueeze(1), rowpeak=peaks)
def st(): return mel(audio, target_mel=tm, rowpeak=peaks)
def rn(): voice.render_audio(ws, out=audio, normalize=False)
def ct(): voice.render_control(ws)

def both():
    main = torch.cuda.current_stream()
    sa.wait_stream(main); sb.wait_stream(main)
    with torch.cuda.stream(sa): pq()
    with torch.cuda.stream(sb): st()
    main.wait_stream(sa); main.wait_stream(sb)

def timeit(name, fn):
    fn(); fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(K): fn()
    g.replay(); torch.cuda.synchronize()
    best = []
    for _ in range(9):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
        best.append(e0.elapsed_time(e1) / K * 1e3)
    best.sort()
    print(f"{name:28s} median {best[4]:8.1f} us   min {best[0]:8.1f} us", flush=True)

timeit("control pass", ct)
timeit("render", rn)
timeit("pqmf", pq)
timeit("stft mel-L1", st)
timeit("pqmf then stft (1 stream)", lambda: (pq(), st()))
timeit("pqmf || stft (2 streams)", both)
