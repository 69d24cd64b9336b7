"""Times the four embedding-gradient scatter-sums of a train step, one by one and batched (diagnostic)."""
import os, sys, json
import torch
sys.path.insert(0, os.getcwd())
from tts_king_amd import ops
DEV = "cuda:0"


def timeit(fn, n=10, reps=5):
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n):
            fn()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        g.replay()
    e1.record(); torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / (n * reps)


B, L, D = 16, 64, 256
rows = B * L
gen = torch.Generator().manual_seed(0)
dx = torch.randn(rows, D, generator=gen).to(torch.bfloat16).to(DEV)
st = json.load(open("pretrained/stats.json"))
pt, et = torch.randn(rows, generator=gen), torch.randn(rows, generator=gen)
pidx = torch.bucketize(pt, torch.linspace(st["pitch"][0], st["pitch"][1], 255)).int().to(DEV)
eidx = torch.bucketize(et, torch.linspace(st["energy"][0], st["energy"][1], 255)).int().to(DEV)
texts = torch.randint(1, 207, (rows,), generator=gen).to(DEV)
spk = torch.randint(0, 65, (B,), generator=gen).to(DEV)
tabs = {"energy": (eidx, torch.zeros(256, D, device=DEV), 1, -1), "pitch": (pidx, torch.zeros(256, D, device=DEV), 1, -1),
        "speaker": (spk, torch.zeros(65, D, device=DEV), L, -1), "word": (texts, torch.zeros(207, D, device=DEV), 1, 0)}
for k, (idx, tab, div, skip) in tabs.items():
    print("%-8s %6.1f us" % (k, timeit(lambda: ops.scatter_sum(dx, idx, tab, idx_div=div, skip_row=skip))))


def batched():
    q = []
    for k, (idx, tab, div, skip) in tabs.items():
        ops.scatter_sum(dx, idx, tab, idx_div=div, skip_row=skip, defer=q)
    ops.flush_finalize(q)


print("batched  %6.1f us" % timeit(batched))
