"""(c1, c2) pair of the C = 128 HiFi-GAN stage at the bench shape (B = 8, 24,576 frames): one pair launch vs two window convs."""
import os, sys
import torch
sys.path.insert(0, os.getcwd())
from tts_king_amd import ops
from tools.debug.gemm_micro_util import timeit
DEV = "cuda:0"
B, ln, C = 8, 24576, 128
x = torch.randn(B, ln, C, device=DEV).half(); xl = torch.randn(B, ln, C, device=DEV).half()
out2 = torch.empty_like(x)
b = torch.randn(C, device=DEV)
tot1 = tot2 = 0.0
for K in (3, 7, 11):
    w = (torch.randn(C, C, K, device=DEV) * (C * K) ** -0.5)
    pack = ops.pack_resblock_weight(w, dtype=torch.float16)
    for dil in (1, 3, 5):
        def two():
            tl = ops.hifi_conv_window(xl, pack, b, K, dil, lrelu_out=True)
            return ops.hifi_conv_window(tl, pack, b, K, 1, R=x, out2=out2)
        t2 = timeit(two)
        t1 = timeit(lambda: ops.hifi_conv_pair(x, pack, b, pack, b, K, dil))
        gf = 2 * 2.0 * B * ln * C * C * K / 1e9
        tot1 += t1; tot2 += t2
        print("K=%2d dil=%d: pair %.1f us (%.0f TF/s useful) | two window convs %.1f us" % (K, dil, t1, gf / t1 * 1e3, t2))
print("total of the 9 pairs: %.1f us vs %.1f us" % (tot1, tot2))

# C = 64 / 32 stages: three pair launches per block vs the six-conv fused kernel
for C, ln in ((64, 49152), (32, 98304)):
    x = torch.randn(B, ln, C, device=DEV).half(); out = torch.empty_like(x)
    b = torch.randn(C, device=DEV)
    for K in (3, 7, 11):
        w = (torch.randn(C, C, K, device=DEV) * (C * K) ** -0.5)
        pack = ops.pack_resblock_weight(w, dtype=torch.float16)
        def three():
            y = x
            for d in (1, 3, 5):
                y = ops.hifi_conv_pair(y, pack, b, pack, b, K, d)
            return y
        t3 = timeit(three, n=5)
        tf = timeit(lambda: ops.hifi_resblock1(x, [pack] * 6, [b] * 6, (1, 3, 5), out, K), n=5)
        gf = 6 * 2.0 * B * ln * C * C * K / 1e9
        print("C=%d K=%2d: three pairs %.1f us (%.0f TF/s useful) | fused six-conv kernel %.1f us" % (C, K, t3, gf / t3 * 1e3, tf))

# C = 256 stage: a pair launch vs the two implicit-GEMM convs it replaces (single launches here; the generator groups three blocks)
C, ln = 256, 3072
x = torch.randn(B, ln, C, device=DEV).half(); xl = torch.randn(B, ln, C, device=DEV).half(); c2 = torch.empty_like(x)
b = torch.randn(C, device=DEV)
tp = tg = 0.0
for K in (3, 7, 11):
    w = (torch.randn(C, C, K, device=DEV) * (C * K) ** -0.5)
    pack = ops.pack_resblock_weight(w, dtype=torch.float16)
    wg = ops.pack_conv_weight(w, dtype=torch.float16)
    for dil in (1, 3, 5):
        def two():
            tl = ops.conv1d(xl, wg, b, dilation=dil, flags=ops.LRELU_OUT, out_slope=0.1)
            return ops.conv1d(tl, wg, b, R=x, C2=c2, flags=ops.C2_LRELU, out_slope=0.1)
        t2 = timeit(two)
        t1 = timeit(lambda: ops.hifi_conv_pair(x, pack, b, pack, b, K, dil))
        gf = 2 * 2.0 * B * ln * C * C * K / 1e9
        tp += t1; tg += t2
        print("C=256 K=%2d dil=%d: pair %.1f us (%.0f TF/s useful) | two implicit-GEMM convs %.1f us" % (K, dil, t1, gf / t1 * 1e3, t2))
print("C=256 total of the 9 pairs: %.1f us vs %.1f us" % (tp, tg))
