"""Where the weights-stationary pair kernel differs from convwin.hip's pair kernel (frames, channels, tiles)."""
import os, sys
import torch
sys.path.insert(0, os.getcwd())
from tts_king_amd import ops
DEV = "cuda:0"
def case(K, dil, B, ln, dt=torch.float16, cap=0):
    C = 64
    g = torch.Generator().manual_seed(K * 1000 + ln + dil)
    x = torch.randn(B, ln, C, generator=g).to(dt).to(DEV)
    w1 = torch.randn(C, C, K, generator=g) * (C * K) ** -0.5
    w2 = torch.randn(C, C, K, generator=g) * (C * K) ** -0.5
    b1, b2 = (0.1 * torch.randn(C, generator=g)).to(DEV), (0.1 * torch.randn(C, generator=g)).to(DEV)
    p1, p2 = ops.pack_resblock_weight(w1.to(DEV), dtype=dt), ops.pack_resblock_weight(w2.to(DEV), dtype=dt)
    want = ops.hifi_conv_pair(x, p1, b1, p2, b2, K, dil)
    got = ops.hifi_conv_pair(x, p1, b1, p2, b2, K, dil, ws=True, max_wgs=cap)
    torch.cuda.synchronize()
    bad = (got != want)
    print("K=%d dil=%d B=%d len=%d cap=%d: %d of %d elements differ" % (K, dil, B, ln, cap, int(bad.sum()), bad.numel()))
    if bad.any():
        fr = bad.any(dim=2)
        for b in range(B):
            idx = fr[b].nonzero().flatten().tolist()
            if idx:
                runs, s = [], idx[0]
                for a, c in zip(idx, idx[1:] + [None]):
                    if c != a + 1:
                        runs.append((s, a)); s = c
                print("  utt %d: frames" % b, runs[:12])
        ch = bad.any(dim=0).any(dim=0).nonzero().flatten().tolist()
        print("  channels:", ch[:70])
        f = fr[0].nonzero().flatten()
        if len(f):
            t = int(f[0])
            print("  first bad frame utt0 %d: got %s want %s" % (t, got[0, t, :8].tolist(), want[0, t, :8].tolist()))
for args in ((3, 1, 1, 192), (3, 1, 1, 384), (3, 1, 2, 700), (7, 3, 1, 192), (3, 1, 1, 100)):
    case(*args)
