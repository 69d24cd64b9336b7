"""Diagnostic: where a step of the weights-stationary pair kernel (csrc/pairws.hip) goes, wave by wave (s_memrealtime stamps, 100 MHz):
stamp 0 step start, 1 end of the wave's tile loop, 2 at the barrier, 3 behind it.  Needs the diagnostic build (`make -C tts_king_amd/csrc stamps`,
TTSK_LIB_PATH=tts_king_amd/libttsk_hip_stamps.so)."""
import ctypes as C, os, sys
import numpy as np
import torch
sys.path.insert(0, os.getcwd())
from tts_king_amd import ops, lib as L
DEV = "cuda:0"
B, ln, Cn = 8, 49152, 64
x = torch.randn(B, ln, Cn, device=DEV).half()
b = torch.randn(Cn, device=DEV)
lib = L.load()
lib.ttsk_hifi_conv_pair_ws_set_stamps.argtypes = [C.c_void_p]
nwg = 256
for K, dil in ((3, 1), (7, 3), (11, 5)):
    w = (torch.randn(Cn, Cn, K, device=DEV) * (Cn * K) ** -0.5)
    pack = ops.pack_resblock_weight(w, dtype=torch.float16)
    for _ in range(3):
        ops.hifi_conv_pair(x, pack, b, pack, b, K, dil, ws=True)
    st = torch.zeros(nwg * 16 * 8 * (4 + 16), dtype=torch.int64, device=DEV)      # [wg][step][wave][4] step stamps, then [wg][step][wave][16] tile stamps
    torch.cuda.synchronize()
    lib.ttsk_hifi_conv_pair_ws_set_stamps(C.c_void_p(st.data_ptr()))
    ops.hifi_conv_pair(x, pack, b, pack, b, K, dil, ws=True)
    torch.cuda.synchronize()
    lib.ttsk_hifi_conv_pair_ws_set_stamps(C.c_void_p(0))
    raw = st.cpu().numpy().astype(np.float64) * 0.01      # us
    s = raw[:nwg * 16 * 8 * 4].reshape(nwg, 16, 8, 4)
    tl = raw[nwg * 16 * 8 * 4:].reshape(nwg, 16, 8, 16)
    t0 = s[:, 0, :, 0].min()
    s = s - t0
    print("K=%d dil=%d: first step starts %.1f..%.1f us after the earliest wave; last barrier passed at %.1f us" % (
        K, dil, s[:, 0, :, 0].min(), s[:, 0, :, 0].max(), s[:, 8, :, 3].max()))
    print("  step | c1 waves (0-3): tile loop, then window store  | c2 waves (4-7): tile loop | wait at barrier c1 / c2 | step length")
    for step in range(9):
        v = s[:, step]                                  # [wg][wave][4]
        c1, c2 = v[:, :4], v[:, 4:]
        print("  %4d | %5.2f  +%5.2f (fh0 %5.2f fh1 %5.2f)           | %5.2f (fh0 %5.2f fh1 %5.2f) | %5.2f / %5.2f | %5.2f" % (
            step, np.median(c1[:, :, 1] - c1[:, :, 0]), np.median(c1[:, :, 2] - c1[:, :, 1]), np.median(c1[:, :2, 1] - c1[:, :2, 0]),
            np.median(c1[:, 2:, 1] - c1[:, 2:, 0]), np.median(c2[:, :, 1] - c2[:, :, 0]), np.median(c2[:, :2, 1] - c2[:, :2, 0]),
            np.median(c2[:, 2:, 1] - c2[:, 2:, 0]), np.median(c1[:, :, 3] - c1[:, :, 2]), np.median(c2[:, :, 3] - c2[:, :, 2]),
            np.median(v[:, :, 3].max(axis=1) - v[:, :, 0].min(axis=1))))
    for step in (0, 3, 8):
        for wv, nt in ((0, 7), (2, 6), (4, 6)):
            d = np.diff(tl[:, step, wv, :nt + 1], axis=1)
            if (tl[:, step, wv, :nt + 1] > 0).all():
                print("    step %d wave %d: per-tile us (median over workgroups): %s" % (step, wv, " ".join("%.2f" % x for x in np.median(d, axis=0))))
