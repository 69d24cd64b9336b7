"""Flash attention forward / backward at the decoder's shape against the batch size: does a second workgroup per CU come for free
(the kernels are bound by one wave per SIMD waiting, not by throughput)?"""
import os, sys
import torch
sys.path.insert(0, os.getcwd())
from tts_king_amd import ops
from tools.debug.gemm_micro_util import timeit
DEV = "cuda:0"
torch.manual_seed(0)
H, d = 2, 256
for S in (423, 64):
    for B in (16, 32, 64):
        qkv = (torch.randn(B * S, 3 * d, device=DEV) * 0.5).bfloat16()
        lens = torch.randint(S * 3 // 4, S + 1, (B,), device=DEV)
        o, lse, o32 = ops.flash_attention_fwd(qkv, lens, B, H, S, True)
        do = torch.randn(B * S, d, device=DEV).bfloat16()
        delta = torch.randn(B * H, S, device=DEV)
        t1 = timeit(lambda: ops.flash_attention_fwd(qkv, lens, B, H, S, True))
        t2 = timeit(lambda: ops.flash_attention_bwd(qkv, o, do, lse, lens, B, H, S, o32=o32, delta=delta))
        print("S=%d B=%d: forward %.1f us | backward %.1f us" % (S, B, t1, t2))
