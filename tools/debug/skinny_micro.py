"""Diagnostic: the PostNet's 512 -> 80 conv (k = 5) and mel_linear (256 -> 80) on each GEMM configuration x split count, under hipGraph
replay (eager timings are host-bound).  usage: python tools/debug/skinny_micro.py"""
import os, sys
import torch
sys.path.insert(0, os.getcwd())
from tts_king_amd import ops
DEV = "cuda:0"
bf = lambda t: t.to(torch.bfloat16)
g = torch.Generator().manual_seed(0)
B, T = 16, 423
def timed(fn, n=50):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        for _ in range(10):
            fn()
    gr.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        gr.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (n * 10)
x5 = bf(torch.randn(B, T, 512, generator=g)).to(DEV)
W5 = bf(torch.randn(80, 5, 512, generator=g) * 0.02).to(DEV)
b5 = torch.randn(80, generator=g).to(DEV)
y = bf(torch.randn(B * T, 256, generator=g)).to(DEV)
Wm = bf(torch.randn(80, 256, generator=g) * 0.06).to(DEV)
for kern in (0, 1, 2, 3):
    for sp in (0, 1, 2, 4, 8):
        try:
            t5 = timed(lambda: ops.conv1d(x5, W5, b5, out_dtype=torch.float32, kernel=kern, splits=sp))
        except Exception as e:
            t5 = float("nan")
        try:
            tm = timed(lambda: ops.linear(y, Wm, b5, out_dtype=torch.float32, kernel=kern, splits=sp))
        except Exception as e:
            tm = float("nan")
        print("kernel %d splits %d: conv 512->80 k=5 %.1f us   linear 256->80 %.1f us" % (kern, sp, t5, tm))
