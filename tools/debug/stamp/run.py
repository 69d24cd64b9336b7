import os, sys, torch
sys.path.insert(0, os.getcwd())
from tts_king_amd import lib
lib.LIB_PATH = os.path.join(os.getcwd(), "tools/debug/stamp/libttsk_stamp.so")
dbg = torch.zeros(8 * 32, dtype=torch.int64, device="cuda:0")
os.environ["RBDBG"] = hex(dbg.data_ptr())
lib._lib = None; lib.load(lib.LIB_PATH)
from tts_king_amd import ops
names = ["other(prologue/epi-tail)", "mma", "wstore+wload", "stage barrier", "epilogue", "conv barrier"]
for C, K, ln in ((64, 11, 49152), (64, 3, 49152), (32, 11, 98304)):
    x = torch.randn(8, ln, C, device="cuda:0").half()
    ws = [ops.pack_resblock_weight(torch.randn(C, C, K, device="cuda:0") * (C * K) ** -0.5) for _ in range(6)]
    bs = [torch.zeros(C, device="cuda:0") for _ in range(6)]
    out = torch.empty_like(x)
    for _ in range(3):
        ops.hifi_resblock1(x, ws, bs, (1, 3, 5), out, K)
    torch.cuda.synchronize()
    d = dbg.cpu().view(8, 32)
    print("C=%d K=%d" % (C, K))
    for w in (0, 3, 7):
        t = d[w, :6].tolist()
        print(" wave", w, "  ".join("%s:%d" % (names[i], t[i]) for i in range(6)), " sum", sum(t))
