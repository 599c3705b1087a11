"""How long the HOST spends inside one hipGraph replay call (vs the GPU time of the replayed work) -- developer tool.
A graph of K empty-ish kernels on one stream, and the same on 4 streams."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
dev = torch.device("cuda:0")
x = torch.zeros(1024, device=dev)
for K in (100, 1000):
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for _ in range(3): x.add_(1.0)
        torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=s):
            for _ in range(K): x.add_(1.0)
    torch.cuda.synchronize()
    for _ in range(3):
        t0 = time.perf_counter(); g.replay(); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"K={K}: replay() call returns after {(t1-t0)*1e3:.2f} ms, all done after {(t2-t0)*1e3:.2f} ms  "
          f"-> {(t1-t0)/K*1e6:.1f} us host per node, {(t2-t0)/K*1e6:.1f} us per node end to end")
