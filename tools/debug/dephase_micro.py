"""Diagnostic: does a C = 128 conv pair get faster when two halves of the batch run on two streams, the second delayed by a
fraction of a workgroup's lifetime (chip-wide de-phasing of the HBM and MFMA phases)?"""
import os, sys
import torch
sys.path.insert(0, os.getcwd())
from tts_king_amd import ops
from tools.debug.gemm_micro_util import timeit
DEV = "cuda:0"
B, ln, C = 8, 24576, 128
x = torch.randn(B, ln, C, device=DEV).half()
b = torch.randn(C, device=DEV)
tiny = torch.randn(1, 96, C, device=DEV).half()
s2 = torch.cuda.Stream()
for K in (3, 7, 11):
    w = (torch.randn(C, C, K, device=DEV) * (C * K) ** -0.5)
    pack = ops.pack_resblock_weight(w, dtype=torch.float16)
    xa, xb = x[:4].contiguous(), x[4:].contiguous()
    t_full = timeit(lambda: ops.hifi_conv_pair(x, pack, b, pack, b, K, 3))
    for ndelay in (0, 1, 2, 4):
        def split():
            cur = torch.cuda.current_stream()
            s2.wait_stream(cur)
            with torch.cuda.stream(s2):
                for _ in range(ndelay):
                    ops.hifi_conv_pair(tiny, pack, b, pack, b, K, 3)
                ops.hifi_conv_pair(xb, pack, b, pack, b, K, 3)
            ops.hifi_conv_pair(xa, pack, b, pack, b, K, 3)
            cur.wait_stream(s2)
        t = timeit(split, n=10)
        print("K=%2d: one launch %.1f us | two half-batch launches on two streams, second after %d tiny launches: %.1f us" % (K, t_full, ndelay, t))
