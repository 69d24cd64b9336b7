"""A few GEMM launches for rocprofv3 --pmc (diagnostic)."""
import os, sys
import torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.getcwd()))
from tts_king_amd import ops
DEV = "cuda:0"; bf = torch.bfloat16
def conv(Bsz, T, Cin, Cout, k):
    x = torch.randn(Bsz, T, Cin, device=DEV).to(bf); w = (torch.randn(Cout, k, Cin, device=DEV) * (Cin * k) ** -0.5).to(bf)
    b = torch.zeros(Cout, device=DEV)
    return lambda **kw: ops.conv1d(x, w, b, **kw)
f1 = conv(8, 24576, 128, 128, 11)      # hifi stage 2
f2 = conv(16, 423, 256, 1024, 9)       # decoder w_1
for kernel in (1, 2):
    for _ in range(3):
        f1(kernel=kernel, splits=1)
    for _ in range(3):
        f2(kernel=kernel, splits=1)
torch.cuda.synchronize()
