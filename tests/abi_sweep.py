"""Run by tests/test_abi_cpu.py in a child process with the AddressSanitizer runtime preloaded and TTSK_LIB_PATH pointing at the
ASan host build (`make -C tts_king_amd/csrc asan`): the argument checking of every entry point include/ttsk.h declares, and the
host-memory work of the planner / grouped-launch table builder, without a GPU.  Nothing is launched: every call either is
host-only or is rejected before its launch.  Prints `abi sweep ok <n>` at the end; an ASan report aborts the process."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tts_king_amd import lib  # noqa: E402

L = lib.load()
protos = lib.declared_prototypes()
n_calls = 0
SKIP = {"ttsk_version", "ttsk_last_error"}
HOST_INT = [k for k, v in protos.items() if k not in SKIP and v and all(a is C.c_int for a in v)]
# (1) host-only size / support queries over a range of arguments, including nonsense ones
for name in HOST_INT:
    fn = getattr(L, name)
    for base in (-3, 0, 1, 7, 32, 64, 80, 256, 512, 1024, 6768, 1 << 20):
        args = [base + 3 * i for i in range(len(protos[name]))]
        fn(*args)
        n_calls += 1
# (2) every entry point that takes pointers, with null pointers and zero sizes: rejected with an error code, nothing launched
for name, argtypes in protos.items():
    if name in SKIP or name in HOST_INT or not argtypes:
        continue
    fn = getattr(L, name)
    zero = [(None if a is C.c_void_p else (0.0 if a in (C.c_float, C.c_double) else 0)) for a in argtypes]
    if name in ("ttsk_gemm", "ttsk_gemm_plan"):
        continue                                   # typed pointers: exercised below
    if name in ("ttsk_scatter_sum_batch", "ttsk_colsum_batch", "ttsk_colsum_finalize_batch", "ttsk_gemm_reduce_batch"):
        rc = fn(None, 0, None)
    else:
        rc = fn(*zero)
    n_calls += 1
    if rc == 0 and not name.endswith(("_bytes", "_nblocks", "_elems", "_rows")):
        raise SystemExit("%s accepted an all-null / all-zero argument list" % name)
# (3) the planner and the grouped-launch table builder on real descriptors (host memory reads and writes)
buf = (C.c_ubyte * (1 << 16))()
base = C.addressof(buf)


def desc(M, N, K, flags=0, taps=0, nz2=1, splits=0):
    d = lib.GemmDesc()
    d.A = d.B = d.C = base                       # never dereferenced on the host
    d.M, d.N, d.K, d.lda, d.ldb, d.ldc = M, N, K, K, K, N
    d.flags, d.alpha, d.nz1, d.nz2, d.taps, d.seg_len, d.splits = flags, 1.0, 1, nz2, taps, (M if taps else 0), splits
    d.tap_shift0, d.tap_dshift, d.b_tap_stride = -(taps // 2), 1, K
    return d


shapes = [(6768, 1024, 256, 0, 9), (6768, 256, 1024, 0, 0), (1024, 768, 256, 0, 0), (6768, 80, 256, 0, 0), (8, 8, 8, 0, 0),
          (1024, 256, 6768, lib.A_TR | lib.B_TR, 0), (512, 512, 6768, lib.A_TR | lib.B_TR, 0), (80, 512, 6768, lib.A_TR | lib.B_TR, 0)]
planned = []
for M, N, K, fl, taps in shapes:
    d = desc(M, N, K, fl, taps)
    k, s, ws = C.c_int32(0), C.c_int32(0), C.c_int64(0)
    rc = L.ttsk_gemm_plan(C.byref(d), C.byref(k), C.byref(s), C.byref(ws))
    n_calls += 1
    assert rc == 0, (M, N, K, L.ttsk_last_error())
    assert s.value >= 1 and ws.value >= 0
    d.kernel, d.splits = k.value, s.value
    planned.append(d)
for group in (planned[5:], planned[5:6] * 50):
    group = [g for g in group if g.splits == 1 or True]
    for g in group:
        g.splits = 1
    n = len(group)
    nbytes = int(L.ttsk_gemm_group_table_bytes(n))
    host = (C.c_ubyte * nbytes)()                 # exactly the advertised size: an overrun is an ASan report
    arr = (lib.GemmDesc * n)(*group)
    total = C.c_int32(0)
    rc = L.ttsk_gemm_group_build(arr, n, host, C.byref(total))
    n_calls += 1
    assert rc == 0 and total.value > 0, L.ttsk_last_error()
if "--overrun" in sys.argv:
    # negative control: a table buffer 64 bytes short of ttsk_gemm_group_table_bytes(n) must be caught by the sanitizer
    small = (C.c_ubyte * (nbytes - 64))()
    L.ttsk_gemm_group_build(arr, n, small, C.byref(total))
    print("overrun not detected")
    sys.exit(0)
d = lib.GemmDesc()
assert L.ttsk_gemm(C.byref(d), None) == -1 and b"null operand" in L.ttsk_last_error()
print("abi sweep ok %d" % n_calls)
