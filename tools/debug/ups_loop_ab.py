"""256 -> 128 stride-8 upsampler at the bench shape (B = 8, T = 3072): win_conv_kernel (one channel group per workgroup) vs ups_loop_kernel, and the
whole generator with either, alternated."""
import os, sys, time
import torch
sys.path.insert(0, os.getcwd())
from tts_king_amd import ops
from tts_king_amd.config import default_config
from tts_king_amd.hifi_bench import build_generator
DEV = "cuda:0"
g = torch.Generator().manual_seed(0)
B, T, Cin, Cout, s = 8, 3072, 256, 128, 8
x = torch.randn(B, T, Cin, generator=g).half().to(DEV)
Wp = (torch.randn(2 * s, Cout, Cin, generator=g) * 0.02).half().to(DEV)
bias = (0.1 * torch.randn(Cout, generator=g)).to(DEV)
pack, brep = ops.hifi_upsample_win_pack(Wp, bias, s)


def t(fn, n=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return 1e6 * (time.perf_counter() - t0) / n


for r in range(3):
    print("win %.1f us | loop %.1f us" % (t(lambda: ops.hifi_upsample_win(x, pack, brep, Cout, s)), t(lambda: ops.hifi_upsample_loop(x, pack, brep, Cout, s))), flush=True)

cfg = default_config()
gens = {}
mel = torch.randn(8, 80, 384, device=DEV)
for name, v in (("win", False), ("loop", True)):
    gen = build_generator(cfg, DEV)
    gen.loop_upsample = v
    gen(mel)
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        y = gen(mel)
    gens[name] = (gr, gen, y)
for r in range(3):
    print(" | ".join("%s %.4f ms" % (k, 1e-3 * t(v[0].replay, 30)) for k, v in gens.items()), flush=True)
