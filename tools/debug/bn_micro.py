"""PostNet BatchNorm apply kernels (forward / backward) at the training shape against the number of statistics partial rows every workgroup
re-sums: how much of the 15-18 us is the re-summation."""
import os, sys
import torch
sys.path.insert(0, os.getcwd())
from tts_king_amd import ops
from tools.debug.gemm_micro_util import timeit
DEV = "cuda:0"
torch.manual_seed(0)
rows, C = 16 * 423, 512
x = torch.randn(rows, C, device=DEV)
gamma, beta = torch.rand(C, device=DEV) + 0.5, torch.randn(C, device=DEV) * 0.1
rm, rv, nbt = torch.zeros(C, device=DEV), torch.ones(C, device=DEV), torch.zeros(1, dtype=torch.int64, device=DEV)
rng = ops.rng_of(ops.optim_state(DEV, seed=3))
dout = torch.randn(rows, C, device=DEV).bfloat16()
for nblk in (112, 28, 7, 1):
    P = torch.rand(nblk, 2 * C, device=DEV) * rows / nblk
    P[:, C:] += P[:, :C] ** 2 / (rows / nblk)
    t1 = timeit(lambda: ops.bn_train(x, rm, rv, nbt, gamma, beta, True, p=0.5, site=40, rng=rng, partials=P, want_keep=True))
    out, mean, rstd, keep = ops.bn_train(x, rm, rv, nbt, gamma, beta, True, p=0.5, site=40, rng=rng, partials=P, want_keep=True)
    t2 = timeit(lambda: ops.bn_bwd(dout, x, mean, rstd, gamma, beta, True, p=0.5, site=40, rng=rng, keep=keep, partials=P))
    print("partial rows %3d: bn_train apply %.1f us | bn_bwd apply %.1f us" % (nblk, t1, t2))
