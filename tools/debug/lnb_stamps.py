"""Diagnostic: where a workgroup of the fused LayerNorm-backward kernel (ttsk_layernorm_bwd_proj) spends its lifetime — s_memrealtime
stamps (100 MHz): [0] start, [1] after the upstream q|k|v product (PRE) and the weight prefetch, [2] LayerNorm rows done, [3] partials
written (barrier), [4] all channel groups projected and stored."""
# Needs the diagnostic build: `make -C tts_king_amd/csrc stamps` and TTSK_LIB_PATH=tts_king_amd/libttsk_hip_stamps.so (the product
# library carries neither the stamp code nor the *_set_stamps hooks).

import ctypes as C, os, sys
import numpy as np
import torch
sys.path.insert(0, os.getcwd())
from tts_king_amd import ops, lib as L
DEV = "cuda:0"
bf = lambda t: t.to(torch.bfloat16)
g = torch.Generator().manual_seed(0)
lib = L.load()
lib.ttsk_layernorm_bwd_proj_set_stamps.argtypes = [C.c_void_p]
for B, S in ((16, 423), (16, 64)):
    d, Fh, rows = 256, 1024, B * S
    z = bf(torch.randn(rows, d, generator=g)).to(DEV)
    mean, rstd = (0.1 * torch.randn(rows, generator=g)).to(DEV), (0.5 + torch.rand(rows, generator=g)).to(DEV)
    gamma = (1 + 0.1 * torch.randn(d, generator=g)).to(DEV)
    lens = torch.randint(S * 3 // 4, S + 1, (B,), generator=g).to(DEV)
    dout = bf(torch.randn(rows, d, generator=g)).to(DEV)
    R = bf(torch.randn(rows, d, generator=g)).to(DEV)
    Wf = bf(torch.randn(d, 1, d, generator=g) * d ** -0.5).to(DEV)
    W2 = bf(torch.randn(d, 1, Fh, generator=g) * Fh ** -0.5).to(DEV)
    Wq = bf(torch.randn(3 * d, 1, d, generator=g) * (3 * d) ** -0.5).to(DEV)
    h = bf(torch.randn(rows, Fh, generator=g)).clamp(min=0).to(DEV)
    dqkv = bf(torch.randn(rows, 3 * d, generator=g)).to(DEV)
    o32 = torch.randn(rows, d, generator=g).to(DEV)
    pf, p2, pq = (torch.empty(w.numel(), dtype=torch.bfloat16, device=DEV) for w in (Wf, W2, Wq))
    ops.win_conv_pack_items([(Wf, pf, True), (W2, p2, True), (Wq, pq, True)])
    rng = ops.rng_of(ops.optim_state(DEV, seed=3))
    delta = torch.empty(B * 2, S, dtype=torch.float32, device=DEV)
    sl = ops.win_conv_split(bf(torch.randn(B, S, 1024, generator=g)).to(DEV), torch.empty(1024 * 9 * 256, dtype=torch.bfloat16, device=DEV).normal_(), 256, 9)
    cases = {
        "<4, PRE> (w_2 dX + q|k|v dX in front)": lambda: ops.layernorm_bwd_proj(None, z, mean, rstd, gamma, p2, Fh, lens, S, p_pre=0.1, site_pre=1, rng=rng, R=R, gate=h, pre=(dqkv, pq)),
        "<4> (w_2 dX, bf16 upstream)": lambda: ops.layernorm_bwd_proj(dout, z, mean, rstd, gamma, p2, Fh, lens, S, p_pre=0.1, site_pre=1, rng=rng, gate=h),
        "<1> (fc dX + delta, 4 slabs + R)": lambda: ops.layernorm_bwd_proj(None, z, mean, rstd, gamma, pf, d, lens, S, p_pre=0.1, site_pre=1, rng=rng, slabs=sl, R=R, delta_o32=o32, delta_out=delta),
    }
    nwg = (rows + 31) // 32
    for name, fn in cases.items():
        for _ in range(3):
            fn()
        st = torch.zeros(nwg * 8, dtype=torch.int64, device=DEV)
        torch.cuda.synchronize()
        lib.ttsk_layernorm_bwd_proj_set_stamps(C.c_void_p(st.data_ptr()))
        fn()
        torch.cuda.synchronize()
        lib.ttsk_layernorm_bwd_proj_set_stamps(C.c_void_p(0))
        s = st.cpu().numpy().reshape(nwg, 8)[:, :5].astype(np.float64) * 0.01
        s -= s[:, 0].min()
        dd = np.diff(s, axis=1)
        print("rows %d %s: %d workgroups, span %.1f us, lifetime mean %.1f (max %.1f); phases mean [PRE+prefetch %.2f | LN rows %.2f | partials %.2f | groups %.2f] us; starts up to %.1f us"
              % (rows, name, nwg, s[:, 4].max(), (s[:, 4] - s[:, 0]).mean(), (s[:, 4] - s[:, 0]).max(), dd[:, 0].mean(), dd[:, 1].mean(), dd[:, 2].mean(), dd[:, 3].mean(), s[:, 0].max()))
