"""Square-GEMM calibration of both tile configurations (compare with the guide's 4096^3 ladder)."""
import os, sys
import torch
sys.path.insert(0, os.getcwd())
from tts_king_amd import ops
from tools.debug.gemm_tune import time_it  # noqa
bf = torch.bfloat16
for n in (2048, 4096, 8192):
    x = torch.randn(n, n, device="cuda:0").to(bf); w = torch.randn(n, n, device="cuda:0").to(bf)
    out = torch.empty(n, n, device="cuda:0", dtype=bf)
    for kernel in (1, 2):
        t = time_it(lambda: ops.linear(x, w, out=out, kernel=kernel, splits=1), n=10)
        print("n=%d kernel=%d %.1f us %.0f TFLOP/s" % (n, kernel, t, 2.0 * n ** 3 / t / 1e6))
    t = time_it(lambda: torch.matmul(x, w.t()), n=10)
    print("n=%d hipBLASLt %.1f us %.0f TFLOP/s" % (n, t, 2.0 * n ** 3 / t / 1e6))
print("FS2 shapes through hipBLASLt (plain GEMM of the same M,N,K; the conv's im2col is not charged)")
for name, M, N, K in [("dec w1 fwd", 6768, 1024, 2304), ("dec w1 dx", 6768, 256, 9216), ("dec w1 dw", 1024, 2304, 6768),
                      ("dec w2 fwd", 6768, 256, 1024), ("dec w2 dx", 6768, 1024, 256), ("dec w2 dw", 256, 1024, 6768),
                      ("qkv fwd", 6768, 768, 256), ("qkv dx", 6768, 256, 768), ("qkv dw", 768, 256, 6768),
                      ("fc", 6768, 256, 256), ("postnet fwd", 6768, 512, 2560), ("postnet dw", 512, 2560, 6768)]:
    x = torch.randn(M, K, device="cuda:0").to(bf); w = torch.randn(N, K, device="cuda:0").to(bf)
    t = time_it(lambda: torch.matmul(x, w.t()), n=20)
    print("%-12s M=%d N=%d K=%d hipBLASLt %.1f us %.0f TFLOP/s" % (name, M, N, K, t, 2.0 * M * N * K / t / 1e6))
