"""Diagnostic: the six decoder w_1 weight gradients (k = 9, 1024 x 256 over 16 x 423 rows) as ONE ttsk_dwconv_batch launch against the
grouped GEMM launch they used to be part of; with and without `lens` (the synthetic batch's own lengths: 6070 of 6768 rows).
usage: python tools/debug/dwconv_micro.py [iters]"""
import os, sys
import torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.getcwd()))
from tts_king_amd import ops
from tts_king_amd.synthetic import make_batch
DEV = "cuda:0"; bf = torch.bfloat16
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 20
B, T = 16, 423
b = make_batch(16, 64, seed=1234)
lens = b[7].to(DEV)
g = torch.Generator(device=DEV).manual_seed(0)
probs = [(torch.randn(B, T, 1024, generator=g, device=DEV).to(bf), torch.randn(B, T, 256, generator=g, device=DEV).to(bf),
          torch.zeros(1024, 9, 256, device=DEV)) for _ in range(6)]
flops = 6 * 2.0 * B * T * 1024 * 256 * 9


def timed(fn, name, fl=flops):
    fn(); fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    print("%-44s %8.1f us  %6.0f TFLOP/s  %5.1f%% of 2.5 PF (algorithmic FLOPs of all %d rows)" % (name, 1e3 * ms, fl / ms / 1e9, 100 * fl / ms / 1e9 / 2500, B * T))


def grouped():
    q = ops.DeferQueue(group_gemms=True)
    for dy, x, dst in probs:
        ops.conv1d_dw(dy, x, dst, k=9, defer=q, accumulate=False)
    ops.flush_deferred(q)


timed(grouped, "grouped GEMM (9 problems per weight)")
timed(lambda: ops.dwconv_batch([(dy, x, dst, None, False) for dy, x, dst in probs]), "dwconv, all rows")
timed(lambda: ops.dwconv_batch([(dy, x, dst, lens, False) for dy, x, dst in probs]), "dwconv, lens (6070 of 6768 rows)")
timed(lambda: ops.dwconv_batch([(dy, x, dst, lens, True) for dy, x, dst in probs]), "dwconv, lens, accumulate")
one = probs[:1]
timed(lambda: ops.dwconv_batch([(dy, x, dst, lens, False) for dy, x, dst in one]), "dwconv, ONE weight (32 workgroups)", flops / 6)
