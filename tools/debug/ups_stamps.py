"""Diagnostic: phase stamps (wc_stamps.py's) of the HiFi-GAN upsamplers on the window-conv kernel at the bench shape (B = 8, 384 mel frames).
Needs `make -C tts_king_amd/csrc stamps` and TTSK_LIB_PATH=tts_king_amd/libttsk_hip_stamps.so."""
import ctypes as C, os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.getcwd())
from tts_king_amd import ops, lib as L
DEV = "cuda:0"
g = torch.Generator().manual_seed(0)
lib = L.load()
lib.ttsk_win_conv_set_stamps.argtypes = [C.c_void_p]
B = 8
for name, T, Cin, Cout, s, TT, cgw in (("ups0 512->256 s8", 384, 512, 256, 8, 96, 256), ("ups1 256->128 s8", 3072, 256, 128, 8, 224, 256),
                                       ("ups2 128->64 s2", 24576, 128, 64, 2, 224, 128)):
    x = torch.randn(B, T, Cin, generator=g).half().to(DEV)
    Wp = (torch.randn(2 * s, Cout, Cin, generator=g) * 0.02).half().to(DEV)
    bias = (0.1 * torch.randn(Cout, generator=g)).to(DEV)
    pack, brep = ops.hifi_upsample_win_pack(Wp, bias, s)
    fn = lambda: ops.hifi_upsample_win(x, pack, brep, Cout, s)
    nwg = ((T + TT - 1) // TT) * B * (s * Cout // cgw)
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    dt = 1e6 * (time.perf_counter() - t0) / 20
    st = torch.zeros(nwg * 24, dtype=torch.int64, device=DEV)
    lib.ttsk_win_conv_set_stamps(C.c_void_p(st.data_ptr()))
    fn()
    torch.cuda.synchronize()
    lib.ttsk_win_conv_set_stamps(C.c_void_p(0))
    raw = st.cpu().numpy().reshape(nwg, 24).astype(np.float64)
    s_ = raw[:, :6] * 0.01
    s_ -= s_[:, 0].min()
    dd = np.diff(s_, axis=1)
    print("%s: %.1f us per launch (20 back to back); %d workgroups, span %.1f us, lifetime mean %.1f (max %.1f); phases mean [window %.2f | barrier %.2f | taps %.2f | "
          "barrier %.2f | staging + stores %.2f] us; starts: median %.1f, max %.1f us"
          % (name, dt, nwg, s_[:, 5].max(), (s_[:, 5] - s_[:, 0]).mean(), (s_[:, 5] - s_[:, 0]).max(), dd[:, 0].mean(), dd[:, 1].mean(), dd[:, 2].mean(),
             dd[:, 3].mean(), dd[:, 4].mean(), np.median(s_[:, 0]), s_[:, 0].max()), flush=True)
