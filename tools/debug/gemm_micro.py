"""Times the train step's GEMM shapes per tile configuration under hipGraph replay (20 back-to-back launches per graph).
usage (GPU box): python tools/debug/gemm_micro.py"""
import os, sys
import torch
sys.path.insert(0, os.getcwd())
from tts_king_amd import ops
DEV = "cuda:0"
bf16 = torch.bfloat16


def timeit(fn, n=20, reps=5):
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n):
            fn()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        g.replay()
    e1.record(); torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / (n * reps)


def rnd(*s):
    return torch.randn(*s, device=DEV).to(bf16)


rows_list = [(6768, 423), (1024, 64)]
print("%-44s %8s" % ("case", "us"))
for M, seg in rows_list:
    x256, x768, x1024 = rnd(M, 256), rnd(M, 768), rnd(M, 1024)
    Wqkv, Wfc, W2, W1 = rnd(768, 256), rnd(256, 256), rnd(256, 1024), rnd(1024, 9, 256)
    Wp = rnd(256, 3, 256)
    b256, b768, b1024 = torch.randn(256, device=DEV), torch.randn(768, device=DEV), torch.randn(1024, device=DEV)
    gamma, beta = torch.ones(256, device=DEV), torch.zeros(256, device=DEV)
    lens = torch.full((M // seg,), seg, dtype=torch.int64, device=DEV)
    for kern in (1, 2, 3):
        for sp in (1, 0):
            tag = "k%d sp%s" % (kern, "auto" if sp == 0 else "1")
            print("M=%d qkv NT N=768 K=256  %s %8.1f" % (M, tag, timeit(lambda: ops.linear(x256, Wqkv, b768, kernel=kern, splits=sp))))
            print("M=%d fc  NT N=256 K=256  %s %8.1f" % (M, tag, timeit(lambda: ops.linear(x256, Wfc, b256, kernel=kern, splits=sp))))
            print("M=%d w2  NT N=256 K=1024 %s %8.1f" % (M, tag, timeit(lambda: ops.linear(x1024, W2, b256, kernel=kern, splits=sp))))
            print("M=%d w2dX BTR N=1024 K=256 gate %s %8.1f" % (M, tag, timeit(lambda: ops.linear_dx(x256, W2, G=x1024, kernel=kern, splits=sp))))
            print("M=%d fcdX BTR N=256 K=256 %s %8.1f" % (M, tag, timeit(lambda: ops.linear_dx(x256, Wfc, kernel=kern, splits=sp))))
            print("M=%d qkvdX BTR N=256 K=768 +R %s %8.1f" % (M, tag, timeit(lambda: ops.linear_dx(x768, Wqkv, R=x256, kernel=kern, splits=sp))))
            print("M=%d k9 conv fwd N=1024 %s %8.1f" % (M, tag, timeit(lambda: ops.conv1d(x256.view(-1, seg, 256), W1, b1024, flags=ops.RELU, kernel=kern, splits=sp))))
            print("M=%d k9 conv dX N=256 %s %8.1f" % (M, tag, timeit(lambda: ops.conv1d_dx(x1024.view(-1, seg, 1024), W1, R=x256.view(-1, seg, 256), kernel=kern, splits=sp))))
            if M == 1024:
                print("M=%d pred conv k3 N=256 %s %8.1f" % (M, tag, timeit(lambda: ops.conv1d(x256.view(-1, seg, 256), Wp, b256, flags=ops.RELU, kernel=kern, splits=sp))))
                print("M=%d pred conv k3 dX %s %8.1f" % (M, tag, timeit(lambda: ops.conv1d_dx(x256.view(-1, seg, 256), Wp, kernel=kern, splits=sp))))
    print("M=%d ln_fwd alone %8.1f" % (M, timeit(lambda: ops.layernorm_fwd(x256, x256, gamma, beta, lens, seg))))
    print("M=%d ln_bwd alone %8.1f" % (M, timeit(lambda: ops.layernorm_bwd(x256, x256, torch.zeros(M, device=DEV), torch.ones(M, device=DEV), gamma, beta, lens, seg))))
    print("M=%d fc+LN fused K=256  %8.1f" % (M, timeit(lambda: ops.gemm_ln_fwd(x256, Wfc, b256, x256, gamma, beta, lens, seg))))
    print("M=%d w2+LN fused K=1024 %8.1f" % (M, timeit(lambda: ops.gemm_ln_fwd(x1024, W2, b256, x256, gamma, beta, lens, seg))))
