"""FFT-block w_1 forward (Conv1d 256 -> 1024, k = 9, ReLU): window kernel vs the implicit-GEMM conv, decoder and encoder shapes."""
import os, sys
import torch
sys.path.insert(0, os.getcwd())
from tts_king_amd import ops
from tools.debug.gemm_micro_util import timeit
DEV = "cuda:0"
for B, S in ((16, 423), (16, 64), (16, 448)):
    x = torch.randn(B, S, 256, device=DEV).bfloat16()
    W = (torch.randn(1024, 9, 256, device=DEV) * 0.02).bfloat16()
    b = torch.randn(1024, device=DEV)
    t1 = timeit(lambda: ops.ffn_conv_fwd(x, W, b))
    pk = ops.ffn_pack_weight(W)
    assert torch.equal(ops.ffn_conv_fwd(x, W, b, packed=pk), ops.ffn_conv_fwd(x, W, b))
    t3 = timeit(lambda: ops.ffn_conv_fwd(x, W, b, packed=pk))
    t4 = timeit(lambda: ops.ffn_pack_weight(W, out=pk))
    print("B=%d S=%d: packed weights %.1f us (pack itself %.1f us)" % (B, S, t3, t4))
    t2 = timeit(lambda: ops.conv1d(x, W, b, flags=ops.RELU))
    gf = 2.0 * B * S * 1024 * 256 * 9 / 1e9
    print("B=%d S=%d: window kernel %.1f us (%.0f TF/s) | implicit GEMM %.1f us (%.0f TF/s)" % (B, S, t1, gf / t1 * 1e3, t2, gf / t2 * 1e3))
