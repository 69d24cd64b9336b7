"""Window conv (C = 128 HiFi-GAN stage) timing at the bench shape: B = 8, 24,576 frames.  the shipped kernel (conv_window2_kernel)."""
import os, sys
import torch
sys.path.insert(0, os.getcwd())
from tts_king_amd import ops
from tools.debug.gemm_micro_util import timeit
DEV = "cuda:0"
B, ln, C = 8, 24576, 128
x = torch.randn(B, ln, C, device=DEV).half(); r = torch.randn(B, ln, C, device=DEV).half()
out2 = torch.empty_like(x)
b = torch.randn(C, device=DEV)
tot = 0.0
for K in (3, 7, 11):
    w = (torch.randn(C, C, K, device=DEV) * (C * K) ** -0.5)
    pack = ops.pack_resblock_weight(w, dtype=torch.float16)
    for dil in (1, 3, 5):
        t1 = timeit(lambda: ops.hifi_conv_window(x, pack, b, K, dil, lrelu_out=True))
        t2 = timeit(lambda: ops.hifi_conv_window(x, pack, b, K, 1, R=r, out2=out2))
        gf = 2.0 * B * ln * C * C * K / 1e9
        tot += t1 + t2
        print("variant %s K=%2d dil=%d: conv+lrelu %.1f us (%.0f TF/s) | conv+R+out2 %.1f us" % (os.environ.get("TTSK_CONVWIN_VARIANT", "2"), K, dil, t1, gf / t1 * 1e3, t2))
print("variant %s total of the 18 convs: %.1f us" % (os.environ.get("TTSK_CONVWIN_VARIANT", "2"), tot))
