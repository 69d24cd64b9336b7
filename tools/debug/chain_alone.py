"""Profiling target for tools/chain_cost.sh: the flash-attention forward / backward kernels ALONE on the chip at the train step's shapes (decoder
B = 16, S = 423; encoder S = 64), 20 launches each back to back — the same kernels the replayed step runs between other launches."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tts_king_amd import ops
DEV = "cuda:0"
torch.manual_seed(0)
H, d, B = 2, 256, 16
for S in (423, 64):
    qkv = (torch.randn(B * S, 3 * d, device=DEV) * 0.5).bfloat16()
    lens = torch.randint(S * 3 // 4, S + 1, (B,), device=DEV)
    lens[0] = S
    o, lse, o32 = ops.flash_attention_fwd(qkv, lens, B, H, S, True)
    do = torch.randn(B * S, d, device=DEV).bfloat16()
    delta = torch.randn(B * H, S, device=DEV)
    for _ in range(20):
        ops.flash_attention_fwd(qkv, lens, B, H, S, True)
    for _ in range(20):
        ops.flash_attention_bwd(qkv, o, do, lse, lens, B, H, S, o32=o32, delta=delta)
torch.cuda.synchronize()
