"""Diagnostic: the step's weight-gradient group (ttsk_gemm_group: gemm2_group_kernel<256, true, true>) alone on the chip, on synthetic
operands of the step's shapes.  usage: python tools/debug/dw_micro.py [all|dec|w1|pn] [iters]   (under rocprofv3 --pmc: iters = 3)
  all: decoder x6 + PostNet + mel_linear + encoder x4 + predictors (what backward queues: 316 GFLOP); dec: the six decoder blocks;
  w1: one decoder w_1 (k = 9, 1024 x 256 over 6768 rows); pn: the PostNet's three 512 x 512 x 5."""
import os, sys, time
import torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.getcwd()))
from tts_king_amd import ops
DEV = "cuda:0"; bf = torch.bfloat16
which = sys.argv[1] if len(sys.argv) > 1 else "all"
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 20
B, T, L = 16, 423, 64
g = torch.Generator(device=DEV).manual_seed(0)
rnd = lambda *s: torch.randn(*s, generator=g, device=DEV).to(bf)
probs = []     # (kind, dy, x, dst, k)


def conv_dw(Bn, S, Cout, Cin, k):
    probs.append(("conv", rnd(Bn, S, Cout), rnd(Bn, S, Cin), torch.zeros(Cout, k, Cin, device=DEV), k))


def lin_dw(rows, Cout, Cin):
    probs.append(("lin", rnd(rows, Cout), rnd(rows, Cin), torch.zeros(Cout, Cin, device=DEV), 1))


def block(Bn, S):
    conv_dw(Bn, S, 256, 1024, 1); conv_dw(Bn, S, 1024, 256, 9); lin_dw(Bn * S, 256, 256); lin_dw(Bn * S, 768, 256)


if which in ("all", "pn"):
    for _ in range(3):
        conv_dw(B, T, 512, 512, 5)
if which == "all":
    conv_dw(B, T, 80, 512, 5); conv_dw(B, T, 512, 80, 5); lin_dw(B * T, 80, 256)
if which in ("all", "dec"):
    for _ in range(6):
        block(B, T)
if which == "all":
    for _ in range(6):
        conv_dw(B, L, 256, 256, 3)
    for _ in range(4):
        block(B, L)
if which == "w1":
    conv_dw(B, T, 1024, 256, 9)
flops = sum(2.0 * p[1].numel() // p[1].shape[-1] * p[1].shape[-1] * p[2].shape[-1] * p[4] for p in probs)


def run():
    q = ops.DeferQueue(group_gemms=True)
    for kind, dy, x, dst, k in probs:
        if kind == "conv":
            ops.conv1d_dw(dy, x, dst, k=k, defer=q, accumulate=False)
        else:
            ops.linear_dw(dy, x, dst, defer=q, accumulate=False)
    n = len(q.group)
    ops.flush_deferred(q)
    return n


n = run(); run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(iters):
    run()
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / iters
print("%s: %d problems (%d grouped), %.1f GFLOP, %.1f us per flush (grouped GEMMs + split-K reducer, host enqueue included) = %.0f TFLOP/s = %.1f%% of 2.5 PF"
      % (which, len(probs), n, flops / 1e9, 1e3 * ms, flops / ms / 1e9, 100 * flops / ms / 1e9 / 2500))
