"""Times kernel 2 on a 4096^3 GEMM and the step's largest conv shapes (TTSK_LIB_PATH selects a diagnostic build of the library)."""
import os, sys
import torch
sys.path.insert(0, os.getcwd())
from tts_king_amd import ops
from tools.debug.gemm_tune import time_it, conv_case  # noqa
bf = torch.bfloat16
for n in (4096,):
    x = torch.randn(n, n, device="cuda:0").to(bf); w = torch.randn(n, n, device="cuda:0").to(bf)
    out = torch.empty(n, n, device="cuda:0", dtype=bf)
    t = time_it(lambda: ops.linear(x, w, out=out, kernel=2, splits=1), n=10)
    print("n=%d kernel=2 %.1f us %.0f TFLOP/s" % (n, t, 2.0 * n ** 3 / t / 1e6))
for name, args, mode, sp in [("dec w1", (16, 423, 256, 1024, 9), "fwd", 1), ("dec w1", (16, 423, 256, 1024, 9), "dx", 4),
                             ("dec w1", (16, 423, 256, 1024, 9), "dw", 3), ("postnet", (16, 423, 512, 512, 5), "fwd", 2)]:
    fn, fl = conv_case(*args, mode)
    t = time_it(lambda: fn(kernel=2, splits=sp))
    print("%s %s splits=%d %.1f us %.0f TFLOP/s" % (name, mode, sp, t, fl / t / 1e6))
