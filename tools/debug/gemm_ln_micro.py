import os, sys
import torch
sys.path.insert(0, os.getcwd())
from tts_king_amd import ops
from tools.debug.gemm_micro_util import timeit
DEV = "cuda:0"
for M, seg in ((6768, 423), (1024, 64)):
    x256, x1024 = torch.randn(M, 256, device=DEV).bfloat16(), torch.randn(M, 1024, device=DEV).bfloat16()
    Wfc, W2 = torch.randn(256, 256, device=DEV).bfloat16(), torch.randn(256, 1024, device=DEV).bfloat16()
    b = torch.randn(256, device=DEV); g = torch.ones(256, device=DEV); be = torch.zeros(256, device=DEV)
    lens = torch.full((M // seg,), seg, dtype=torch.int64, device=DEV)
    print("variant %s M=%d fc+LN %.1f us  w2+LN %.1f us" % (os.environ.get("TTSK_GEMM_LN_VARIANT", "default"), M,
          timeit(lambda: ops.gemm_ln_fwd(x256, Wfc, b, x256, g, be, lens, seg)), timeit(lambda: ops.gemm_ln_fwd(x1024, W2, b, x256, g, be, lens, seg))))
