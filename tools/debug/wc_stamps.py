"""Diagnostic: where a workgroup of the window conv kernel (ttsk_win_conv / ttsk_win_conv_split / ttsk_ffn_conv_fwd) spends its
lifetime — s_memrealtime stamps (100 MHz): [0] start, [1] window in LDS (own part), [2] barrier passed, [3] tap loop done,
[4] barrier passed, [5] outputs staged and stored."""
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
lib.ttsk_win_conv_set_stamps.argtypes = [C.c_void_p]
B, S, d, Fh = 16, 423, 256, 1024
x = bf(torch.randn(B, S, d, generator=g)).to(DEV)
dh = bf(torch.randn(B, S, Fh, generator=g)).to(DEV)
W1 = bf(torch.randn(Fh, 9, d, generator=g) * (9 * d) ** -0.5).to(DEV)
b1 = (0.1 * torch.randn(Fh, generator=g)).to(DEV)
pk, pkT = (torch.empty(W1.numel(), dtype=torch.bfloat16, device=DEV) for _ in range(2))
ops.win_conv_pack_items([(W1, pk, False), (W1, pkT, True)])
cases = {
    "w_1 forward (256 -> 1024, k = 9, ReLU)": (lambda: ops.ffn_conv_fwd(x, W1, b1, relu=True, packed=pk), 256),
    "w_1 input gradient (4 slices of 256 -> 256, k = 9, fp32 slabs)": (lambda: ops.win_conv_split(dh, pkT, d, 9), 256),
}
xp = bf(torch.randn(B, S, 512, generator=g)).to(DEV)
Wp = bf(torch.randn(512, 5, 512, generator=g) * 0.02).to(DEV)
bp = (0.1 * torch.randn(512, generator=g)).to(DEV)
pp, ppT = (torch.empty(Wp.numel(), dtype=torch.bfloat16, device=DEV) for _ in range(2))
ops.win_conv_pack_items([(Wp, pp, False), (Wp, ppT, True)])
npost = 256 if os.environ.get("WC_POST") else 224
cases["PostNet forward (512 -> 512, k = 5, fp32 out)"] = (lambda: ops.win_conv(xp, pp, 512, 5, bias=bp, out_dtype=torch.float32), npost)
cases["PostNet input gradient (512 -> 512, k = 5)"] = (lambda: ops.win_conv(xp, ppT, 512, 5), npost)
for name, (fn, nwg) in cases.items():
    for _ in range(int(os.environ.get("WARM", "3"))):
        fn()
    st = torch.zeros(nwg * 24, dtype=torch.int64, device=DEV)
    torch.cuda.synchronize()
    lib.ttsk_win_conv_set_stamps(C.c_void_p(st.data_ptr()))
    fn()
    torch.cuda.synchronize()
    lib.ttsk_win_conv_set_stamps(C.c_void_p(0))
    rawi = st.cpu().numpy().reshape(nwg, 24)
    hw = rawi[:, 16:24]
    print("   SIMD of waves 0..7 (first workgroups): " + " | ".join(" ".join(str(int(v >> 4) & 3) for v in hw[k]) for k in range(3)) + "; wave slots: " + " ".join(str(int(v) & 15) for v in hw[0]))
    raw = rawi.astype(np.float64)
    wv = (raw[:, 8:16] - raw[:, 2:3]) * 0.01
    print("   tap loop per wave (us after the barrier, mean over workgroups): " + " ".join("%.1f" % v for v in wv.mean(axis=0)))
    clk = np.median((raw[:, 7] - raw[:, 6]) / (raw[:, 3] - raw[:, 2])) * 0.1        # GHz: shader-clock ticks per 100 MHz tick over the tap loop
    print("   in-kernel clock over the tap loop: %.2f GHz (median over workgroups)" % clk)
    s = raw[:, :6] * 0.01
    s -= s[:, 0].min()
    dd = np.diff(s, axis=1)
    print("%s: %d workgroups, span %.1f us, lifetime mean %.1f (max %.1f); phases mean [window %.2f | barrier %.2f | taps %.2f | barrier %.2f | staging + stores %.2f] us; starts up to %.1f us"
          % (name, nwg, s[:, 5].max(), (s[:, 5] - s[:, 0]).mean(), (s[:, 5] - s[:, 0]).max(), dd[:, 0].mean(), dd[:, 1].mean(), dd[:, 2].mean(), dd[:, 3].mean(), dd[:, 4].mean(), s[:, 0].max()))
