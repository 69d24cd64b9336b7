"""Diagnostic: where a conv-pair workgroup's lifetime goes (s_memrealtime stamps, 100 MHz): load, barrier, c1, t write + barrier,
c2, epilogue staging + barrier, store issue; and how the 2,048 workgroups of a launch are spread over time."""
# Needs the diagnostic build: `make -C tts_king_amd/csrc stamps` and TTSK_LIB_PATH=tts_king_amd/libttsk_hip_stamps.so (the product
# library carries neither the stamp code nor the *_set_stamps hooks).

import ctypes as C, os, sys
import numpy as np
import torch
sys.path.insert(0, os.getcwd())
from tts_king_amd import ops, lib as L
DEV = "cuda:0"
B, ln, Cn = 8, 24576, 128
x = torch.randn(B, ln, Cn, device=DEV).half()
b = torch.randn(Cn, device=DEV)
lib = L.load()
lib.ttsk_hifi_conv_pair_set_stamps.argtypes = [C.c_void_p]
nwg = B * ((ln + 95) // 96)
for K in (7, 11):
    w = (torch.randn(Cn, Cn, K, device=DEV) * (Cn * K) ** -0.5)
    pack = ops.pack_resblock_weight(w, dtype=torch.float16)
    for _ in range(3):
        ops.hifi_conv_pair(x, pack, b, pack, b, K, 3)
    st = torch.zeros(nwg * 8, dtype=torch.int64, device=DEV)
    torch.cuda.synchronize()
    lib.ttsk_hifi_conv_pair_set_stamps(C.c_void_p(st.data_ptr()))
    ops.hifi_conv_pair(x, pack, b, pack, b, K, 3)
    torch.cuda.synchronize()
    lib.ttsk_hifi_conv_pair_set_stamps(C.c_void_p(0))
    s = st.cpu().numpy().reshape(nwg, 8).astype(np.float64) * 0.01      # us
    t0 = s[:, 0].min()
    s -= t0
    names = ["load x window -> LDS", "barrier", "c1 taps", "t write + barrier", "c2 taps", "epilogue staging + barrier", "store issue"]
    d = np.diff(s, axis=1)
    print("K=%d: launch span %.1f us (first start -> last end); workgroup lifetime mean %.1f us (min %.1f, max %.1f)" % (
        K, s[:, 7].max(), (s[:, 7] - s[:, 0]).mean(), (s[:, 7] - s[:, 0]).min(), (s[:, 7] - s[:, 0]).max()))
    for i, n in enumerate(names):
        print("   %-30s mean %6.2f us   p10 %6.2f   p90 %6.2f" % (n, d[:, i].mean(), np.percentile(d[:, i], 10), np.percentile(d[:, i], 90)))
    starts = np.sort(s[:, 0])
    print("   start times: 25%% %.1f  50%% %.1f  75%% %.1f  100%% %.1f us; workgroups alive on average: %.0f of 512 slots" % (
        starts[nwg // 4], starts[nwg // 2], starts[3 * nwg // 4], starts[-1], (s[:, 7] - s[:, 0]).sum() / s[:, 7].max()))
