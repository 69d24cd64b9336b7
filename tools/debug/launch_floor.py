"""Cost of one trivial kernel inside a replayed hipGraph (diagnostic): chain of N dependent tiny launches."""
import os, sys, time
import torch
sys.path.insert(0, os.getcwd())
from tts_king_amd import ops
a = torch.zeros(256, device="cuda:0"); b = torch.ones(256, device="cuda:0")
for N in (100, 500):
    g = torch.cuda.CUDAGraph()
    ops.add_f32(a, b, out=a)
    torch.cuda.synchronize()
    with torch.cuda.graph(g):
        for _ in range(N):
            ops.add_f32(a, b, out=a)
    for _ in range(3): g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20): g.replay()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 20
    print("N=%d trivial kernels per graph: %.1f us per replay = %.2f us per kernel" % (N, dt * 1e6, dt * 1e6 / N))
