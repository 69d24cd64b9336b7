"""C = 64 pairs at the bench shape (B = 8, 49,152 frames; `128`: the C = 128 stage's k = 3 pairs, 24,576 frames): the weights-stationary persistent kernel (csrc/pairws.hip) against convwin.hip's pair
kernel, launch by launch, and the whole MRF stage (9 pairs | fused k = 3 block + 6 pairs).  usage (GPU box): python tools/debug/pairws_micro.py [128]"""
import os, sys
import torch
sys.path.insert(0, os.getcwd())
from tts_king_amd import ops
from tools.debug.gemm_micro_util import timeit
DEV = "cuda:0"
C = int(sys.argv[1]) if len(sys.argv) > 1 else 64
B, ln = 8, 49152 * 64 // C
KS = (3, 7, 11) if C == 64 else (3,)
x = torch.randn(B, ln, C, device=DEV).half()
out = torch.randn(B, ln, C, device=DEV).half()
b = torch.randn(C, device=DEV)
tot = {"old": 0.0, "ws": 0.0}
for K in KS:
    w = (torch.randn(C, C, K, device=DEV) * (C * K) ** -0.5)
    pack = ops.pack_resblock_weight(w, dtype=torch.float16)
    for dil in (1, 3, 5):
        for mode in ((0, 2) if dil == 5 else (0,)):
            kw = dict(out=out, mode=mode, scale=1 / 3.0, final_slope=0.1) if mode else {}
            t_old = timeit(lambda: ops.hifi_conv_pair(x, pack, b, pack, b, K, dil, **kw))
            t_ws = timeit(lambda: ops.hifi_conv_pair(x, pack, b, pack, b, K, dil, ws=True, **kw))
            gf = 2 * 2.0 * B * ln * C * C * K / 1e9
            if mode == 0:
                tot["old"] += t_old; tot["ws"] += t_ws
            print("K=%2d dil=%d mode=%d: pair %.1f us (%.0f TF/s) | weights-stationary %.1f us (%.0f TF/s)" % (K, dil, mode, t_old, gf / t_old * 1e3, t_ws, gf / t_ws * 1e3))
print("%d pairs (mode 0): %.1f us | %.1f us" % (3 * len(KS), tot["old"], tot["ws"]))
for cap in ((64, 128, 192, 224, 256) if C == 64 else ()):
    pack = ops.pack_resblock_weight(torch.randn(C, C, 7, device=DEV) * (C * 7) ** -0.5, dtype=torch.float16)
    print("K=7 dil=3 grid cap %3d: %.1f us" % (cap, timeit(lambda: ops.hifi_conv_pair(x, pack, b, pack, b, 7, 3, ws=True, max_wgs=cap))))
