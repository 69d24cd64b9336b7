"""Times both tile configurations of ttsk_gemm over split-K factors on the shapes of the FS2 step / HiFi-GAN (diagnostic;
calibrates the cost model in csrc/gemm.hip:make_plan)."""
import os, sys
import torch
sys.path.insert(0, os.getcwd())
from tts_king_amd import ops
DEV = "cuda:0"
bf = torch.bfloat16

def time_it(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n

def conv_case(Bsz, T, Cin, Cout, k, mode):
    x = torch.randn(Bsz, T, Cin, device=DEV).to(bf); w = (torch.randn(Cout, k, Cin, device=DEV) * (Cin * k) ** -0.5).to(bf)
    dy = torch.randn(Bsz, T, Cout, device=DEV).to(bf); b = torch.zeros(Cout, device=DEV)
    dw = torch.zeros(Cout, k, Cin, device=DEV)
    if mode == "fwd": return lambda **kw: ops.conv1d(x, w, b, **kw), 2.0 * Bsz * T * Cin * Cout * k
    if mode == "dx": return lambda **kw: ops.conv1d_dx(dy, w, **kw), 2.0 * Bsz * T * Cin * Cout * k
    return lambda **kw: ops.conv1d_dw(dy, x, dw, k=k, **kw), 2.0 * Bsz * T * Cin * Cout * k

cases = []
for name, args in [("dec w1", (16, 423, 256, 1024, 9)), ("dec w2", (16, 423, 1024, 256, 1)), ("postnet", (16, 423, 512, 512, 5)),
                   ("enc w1", (16, 64, 256, 1024, 9)), ("enc w2", (16, 64, 1024, 256, 1)), ("dec qkv", (16, 423, 256, 768, 1)),
                   ("dec fc", (16, 423, 256, 256, 1)), ("pred k3", (16, 64, 256, 256, 3)),
                   ("hifi s2 k11", (8, 24576, 128, 128, 11)), ("hifi s1 k11", (8, 3072, 256, 256, 11)), ("hifi s2 k3", (8, 24576, 128, 128, 3))]:
    for mode in ("fwd", "dx", "dw"):
        if name.startswith("hifi") and mode != "fwd": continue
        cases.append((name + " " + mode,) + conv_case(*args, mode))

if __name__ != "__main__": cases = []
print("%-18s %-3s %s" % ("case", "k", "  ".join("sp=%-2d" % s for s in (1, 2, 3, 4, 6, 8, 12, 16, 24, 32))) + "   auto")
for name, fn, fl in cases:
    for kernel in (1, 2):
        row = []
        for sp in (1, 2, 3, 4, 6, 8, 12, 16, 24, 32):
            try:
                row.append("%5.1f" % time_it(lambda: fn(kernel=kernel, splits=sp)))
            except Exception as e:
                row.append("  err")
        print("%-18s %-3d %s" % (name, kernel, "  ".join(row)))
    t = time_it(lambda: fn())
    print("%-18s auto %.1f us  %.0f TFLOP/s" % (name, t, fl / t / 1e6))
