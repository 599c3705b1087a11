"""Cycle stamps of workgroup 0 of the MFMA PQMF kernel (library built with -DPQM_STAMPS; stamps overwrite z) -- developer tool."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from inverse_audio_synthesis_amd.pqmf import PQMF

dev = torch.device("cuda:0")
B, T, N = int(os.environ.get("B", 128)), int(os.environ.get("T", 176400)), int(os.environ.get("N", 3))
m = PQMF(N).to(dev)
x = torch.randn(B, 1, T, device=dev)
for _ in range(3):
    z = m(x)
torch.cuda.synchronize()
st = z.flatten()[:4 * 6 * 5 * 2].view(torch.int64).cpu().view(4, 6, 5)
names = ["compute", "epilogue->lds", "store", "barrier2", "stage+barrier1 (next)"]
for w in range(4):
    print(f"wave {w}")
    for n in range(6):
        s = st[w, n]
        d = [int(s[i + 1] - s[i]) for i in range(4)]
        nxt = int(st[w, n + 1, 0] - s[4]) if n < 5 else -1
        print(f"  tile {n}: compute {d[0]:6d}  epi {d[1]:6d}  store {d[2]:6d}  barrier2 {d[3]:6d}  stage+barrier1 {nxt:6d}")
