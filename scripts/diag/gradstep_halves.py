"""Timing experiment only (NOT the product path, NOT the reference's loss: every part normalises its spectral convergence over
its own rows): the configs[4] gradient step at B = 64 as H independent parts of B / H voices, each part a captured graph of
its own, the graphs replayed side by side on H streams -- how much of the step's 0.39 ms of chain latency hides behind the other parts'
transform work.   H=1|2|4  INNER=1 (each part's loss branches on side streams of their own, as shipped) | 0 (a part is one stream)
usage (GPU box):  H=2 INNER=0 python3 scripts/diag/gradstep_halves.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import inverse_audio_synthesis_amd  # noqa: F401
from inverse_audio_synthesis_amd.pqmf import PQMF
from inverse_audio_synthesis_amd.spectral import MultiResolutionSTFTLoss, ParallelLossSum, SubbandL1
from inverse_audio_synthesis_amd.voice import SynthConfig, Voice

H = int(os.environ.get("H", "2"))
INNER = os.environ.get("INNER", "1") == "1"
STEPS = int(os.environ.get("STEPS", "20"))
B = 64 // H
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
one = torch.ones((), dtype=torch.float32, device=dev)
parts = []
for h in range(H):
    cfg = SynthConfig(batch_size=B, sample_rate=44100, buffer_size_seconds=4.0, reproducible=False)
    voice = Voice(cfg).to(dev)
    gram = PQMF(N=64).to(dev)
    mr = MultiResolutionSTFTLoss().to(dev)
    sub = SubbandL1(gram)
    both = ParallelLossSum(mr, sub)
    both.parallel = mr.parallel = INNER
    params = torch.rand(B, 78, generator=torch.Generator().manual_seed(1000 + h)).to(dev).requires_grad_(True)
    tgt = voice.render(torch.rand(B, 78, generator=torch.Generator().manual_seed(2000 + h)).to(dev)).clone()
    parts.append(dict(voice=voice, both=both, params=params, tb=sub.target(tgt), tm=mr.target(tgt),
                      stream=torch.cuda.Stream(dev) if H > 1 else None))
out = {}
streams = [torch.cuda.Stream(dev) for _ in parts]


def step_part(h):
    p = parts[h]
    a = p["voice"].render(p["params"])
    loss = p["both"](a, [dict(targets=p["tm"]), dict(target_bands=p["tb"])])
    (g,) = torch.autograd.grad(loss, p["params"], one)
    out[h] = (loss.detach(), g)


# one graph per part (STEPS steps of that part, captured exactly like bench.py's gradient step), the parts' graphs replayed
# side by side on streams of their own: one graph over forked streams crashed hipStreamEndCapture (two tries, not pursued)
graphs = []
if os.environ.get("MODE", "multi") == "one":
    # ONE graph, the parts as branches forked from the capturing stream (crashes hipStreamEndCapture with the shipped
    # BackwardPrelude; IAS_VOICE_PRELUDE=0 to try without its shared side stream)
    dummy = torch.zeros(64, device=dev)

    def step_all():
        cur = torch.cuda.current_stream(dev)
        dummy.zero_()
        for h in range(H):
            streams[h].wait_stream(cur)
            with torch.cuda.stream(streams[h]):
                step_part(h)
        for h in range(H):
            cur.wait_stream(streams[h])

    for _ in range(3):
        step_all()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(STEPS):
            step_all()
    graphs.append(g)
    streams = [torch.cuda.current_stream(dev)]
else:
    for h in range(H):
        for _ in range(3):
            step_part(h)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for _ in range(STEPS):
                step_part(h)
        graphs.append(g)
torch.cuda.synchronize()


def replay_all():
    for h, g in enumerate(graphs):
        with torch.cuda.stream(streams[h]):
            g.replay()


replay_all()
torch.cuda.synchronize()
ts = []
for _ in range(15):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    replay_all()
    torch.cuda.synchronize()
    ts.append((time.perf_counter() - t0) / STEPS * 1e3)
ts.sort()
print(f"H={H} INNER={int(INNER)} B/part={B}: {ts[len(ts) // 2]:.4f} ms/step (min {ts[0]:.4f}), losses "
      + " ".join(f"{float(out[h][0]):.4f}" for h in range(H)))
