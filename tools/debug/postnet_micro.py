import os, sys
import torch
sys.path.insert(0, os.getcwd())
from tts_king_amd import ops
from tools.debug.gemm_micro_util import timeit
DEV = "cuda:0"
B, T = 16, 423
x512 = torch.randn(B, T, 512, device=DEV).bfloat16(); x80 = torch.randn(B, T, 80, device=DEV).bfloat16()
W = torch.randn(512, 5, 512, device=DEV).bfloat16(); W0 = torch.randn(512, 5, 80, device=DEV).bfloat16(); W4 = torch.randn(80, 5, 512, device=DEV).bfloat16()
b512 = torch.randn(512, device=DEV); b80 = torch.randn(80, device=DEV)
for kern in (1, 2, 3):
    for sp in (1, 2, 3, 0):
        try:
            t1 = timeit(lambda: ops.conv1d(x512, W, b512, out_dtype=torch.float32, kernel=kern, splits=sp))
            t2 = timeit(lambda: ops.conv1d_dx(x512, W, kernel=kern, splits=sp))
            t3 = timeit(lambda: ops.conv1d(x80, W0, b512, out_dtype=torch.float32, kernel=kern, splits=sp))
            t4 = timeit(lambda: ops.conv1d(x512, W4, b80, out_dtype=torch.float32, kernel=kern, splits=sp))
            print("kernel %d splits %s: conv 512->512 fwd %.1f dX %.1f | 80->512 %.1f | 512->80 %.1f us" % (kern, sp or "auto", t1, t2, t3, t4))
        except Exception as e:
            print("kernel", kern, "splits", sp, "failed:", str(e)[:80])
