"""K-step time of both tile configurations against the number of workgroups on the chip (latency- or bandwidth-bound?)."""
import os, sys
import torch
sys.path.insert(0, os.getcwd())
from tts_king_amd import ops
from tools.debug.gemm_tune import time_it  # noqa
bf = torch.bfloat16
K = 8192
for kernel, bm, bn in ((1, 128, 128), (2, 256, 128)):
    for nwg in (1, 64, 256, 1024):
        M = bm * nwg
        x = torch.randn(M, K, device="cuda:0").to(bf); w = torch.randn(bn, K, device="cuda:0").to(bf)
        out = torch.empty(M, bn, device="cuda:0", dtype=bf)
        t = time_it(lambda: ops.linear(x, w, out=out, kernel=kernel, splits=1), n=10)
        print("kernel=%d wgs=%4d  %.1f us  %.3f us/K-step  %.0f TFLOP/s" % (kernel, nwg, t, t / (K / 64) / max(1, nwg / 256), 2.0 * M * bn * K / t / 1e6))
