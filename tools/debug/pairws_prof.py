"""Profiling target: the C = 64 pair kernels (old | weights-stationary) a few launches each at the bench shape, for rocprofv3 passes
(tools/pmc_kernel.sh tools/debug/pairws_prof.py)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tts_king_amd import ops
DEV = "cuda:0"
B, ln, C = 8, 49152, 64
x = torch.randn(B, ln, C, device=DEV).half()
b = torch.randn(C, device=DEV)
for K, dil in ((3, 1), (7, 3), (11, 5)):
    w = (torch.randn(C, C, K, device=DEV) * (C * K) ** -0.5)
    pack = ops.pack_resblock_weight(w, dtype=torch.float16)
    for _ in range(4):
        ops.hifi_conv_pair(x, pack, b, pack, b, K, dil)
        ops.hifi_conv_pair(x, pack, b, pack, b, K, dil, ws=True)
torch.cuda.synchronize()
