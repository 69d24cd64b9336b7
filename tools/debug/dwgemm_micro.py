"""Diagnostic: the decoder's k = 1 weight gradients (w_2, q|k|v, fc of six blocks) and the PostNet's three 512 x 512 x 5, as
ttsk_dwgemm_batch launches (with the slab reducer) against the grouped GEMM launch.  usage: python tools/debug/dwgemm_micro.py [iters]"""
import os, sys
import torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.getcwd()))
from tts_king_amd import ops, lib as L
from tts_king_amd.synthetic import make_batch
DEV = "cuda:0"; bf = torch.bfloat16
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 20
B, T = 16, 423
lens = make_batch(16, 64, seed=1234)[7].to(DEV)
g = torch.Generator(device=DEV).manual_seed(0)
rnd = lambda *s: torch.randn(*s, generator=g, device=DEV).to(bf)
probs = []
for _ in range(6):
    probs += [(rnd(B, T, 256), rnd(B, T, 1024), torch.zeros(256, 1, 1024, device=DEV), lens),
              (rnd(B, T, 768), rnd(B, T, 256), torch.zeros(768, 1, 256, device=DEV), lens),
              (rnd(B, T, 256), rnd(B, T, 256), torch.zeros(256, 1, 256, device=DEV), lens)]
pn = [(rnd(B, T, 512), rnd(B, T, 512), torch.zeros(512, 5, 512, device=DEV), None) for _ in range(3)]
flops = lambda ps: sum(2.0 * B * T * p[0].shape[2] * p[1].shape[2] * p[2].shape[1] for p in ps)


def timed(fn, name, fl):
    fn(); fn()
    torch.cuda.synchronize()
    g_ = torch.cuda.CUDAGraph()          # replayed graph: device time, not the host's enqueue time
    with torch.cuda.graph(g_):
        fn()
    g_.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        g_.replay()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    print("%-58s %8.1f us  %6.0f TFLOP/s  %5.1f%% of 2.5 PF" % (name, 1e3 * ms, fl / ms / 1e9, 100 * fl / ms / 1e9 / 2500))


def grouped(ps):
    q = ops.DeferQueue(group_gemms=True)
    for dy, x, dst, _ in ps:
        if dst.shape[1] == 1:
            ops.linear_dw(dy.view(B * T, -1), x.view(B * T, -1), dst.view(dst.shape[0], dst.shape[2]), defer=q, accumulate=False)
        else:
            ops.conv1d_dw(dy, x, dst, k=dst.shape[1], defer=q, accumulate=False)
    ops.flush_deferred(q)


def dwg(ps, splits):
    red = ops.dwgemm_batch([(dy, x, dst, ln, False, splits) for dy, x, dst, ln in ps])
    if red:
        arr = (L.ReduceItem * len(red))(*[r for r, _ in red])
        L.check(L.load().ttsk_gemm_reduce_batch(arr, len(red), torch.cuda.current_stream().cuda_stream), "reduce")


allp = probs + pn
timed(lambda: grouped(allp), "grouped GEMM: 18 decoder k=1 + 3 PostNet k=5", flops(allp))
for sp in (1, 2, 4, 8):
    timed(lambda: dwg(allp, sp), "dwgemm + reducer, %d split(s)" % sp, flops(allp))
timed(lambda: grouped(probs), "grouped GEMM: decoder k=1 only", flops(probs))
for sp in (2, 4, 8):
    timed(lambda: dwg(probs, sp), "dwgemm decoder k=1 only, %d splits" % sp, flops(probs))
timed(lambda: grouped(pn), "grouped GEMM: PostNet only", flops(pn))
for sp in (1, 2, 4):
    timed(lambda: dwg(pn, sp), "dwgemm PostNet only, %d split(s)" % sp, flops(pn))
