"""Diagnostic: where a workgroup of win_ln_kernel (fc / w_2 + dropout + residual + LayerNorm [+ the next q|k|v projection]) spends its
lifetime — s_memrealtime stamps (100 MHz): [0] start, [1] rows of A in LDS (own part), [2] barrier passed, [3] K loop done, [4] barrier
passed, [5] fp32 tile + LayerNorm rows done, [6] projection done.  Needs `make -C tts_king_amd/csrc stamps` and
TTSK_LIB_PATH=tts_king_amd/libttsk_hip_stamps.so.  usage: python tools/debug/wl_stamps.py"""
import ctypes as C, os, sys
import numpy as np
import torch
sys.path.insert(0, os.getcwd())
from tts_king_amd import ops, lib as L
DEV = "cuda:0"
bf = lambda t: t.to(torch.bfloat16)
g = torch.Generator().manual_seed(0)
lib = L.load()
lib.ttsk_win_ln_set_stamps.argtypes = [C.c_void_p]
d = 256
for name, B, S, K, proj in (("encoder w_2 + LN + q|k|v", 16, 64, 1024, True), ("decoder w_2 + LN + q|k|v", 16, 423, 1024, True),
                            ("encoder w_2 + LN", 16, 64, 1024, False), ("decoder fc + LN", 16, 423, 256, False), ("encoder fc + LN", 16, 64, 256, False)):
    rows = B * S
    x = bf(torch.randn(rows, K, generator=g)).to(DEV)
    W = bf(torch.randn(d, 1, K, generator=g) * K ** -0.5).to(DEV)
    Wq = bf(torch.randn(3 * d, 1, d, generator=g) * d ** -0.5).to(DEV)
    bias, bq = (0.1 * torch.randn(d, generator=g)).to(DEV), (0.1 * torch.randn(3 * d, generator=g)).to(DEV)
    res = bf(torch.randn(rows, d, generator=g)).to(DEV)
    gamma, beta = (1 + 0.1 * torch.randn(d, generator=g)).to(DEV), (0.1 * torch.randn(d, generator=g)).to(DEV)
    pw, pq = (torch.empty(w.numel(), dtype=torch.bfloat16, device=DEV) for w in (W, Wq))
    ops.win_conv_pack_items([(W, pw, False), (Wq, pq, False)])
    fn = lambda: ops.win_ln_fwd(x, pw, bias, res, gamma, beta, proj=(pq, bq) if proj else None)
    nwg = (rows + 31) // 32
    for _ in range(3):
        fn()
    st = torch.zeros(nwg * 8, dtype=torch.int64, device=DEV)
    torch.cuda.synchronize()
    lib.ttsk_win_ln_set_stamps(C.c_void_p(st.data_ptr()))
    fn()
    torch.cuda.synchronize()
    lib.ttsk_win_ln_set_stamps(C.c_void_p(0))
    s = st.cpu().numpy().reshape(nwg, 8)[:, :7].astype(np.float64) * 0.01
    s -= s[:, 0].min()
    dd = np.diff(s, axis=1)
    print("%s: %d workgroups, span %.1f us, lifetime mean %.1f (max %.1f); phases mean [A rows %.2f | barrier %.2f | K loop %.2f | barrier %.2f | tile + LayerNorm %.2f | projection %.2f] us; starts up to %.1f us"
          % (name, nwg, s[:, 6].max(), (s[:, 6] - s[:, 0]).mean(), (s[:, 6] - s[:, 0]).max(), dd[:, 0].mean(), dd[:, 1].mean(), dd[:, 2].mean(), dd[:, 3].mean(), dd[:, 4].mean(), dd[:, 5].mean(), s[:, 0].max()))
