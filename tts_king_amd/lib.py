"""ctypes binding of libttsk_hip.so (include/ttsk.h).

The product path has no CPU fallback: if the shared library is missing or was built without a symbol
declared in the header, importing/using the kernels raises immediately.
"""
import ctypes as C
import os
import re

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("TTSK_LIB_PATH") or os.path.join(_HERE, "libttsk_hip.so")   # the override is for diagnostic builds (switches.py)
HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "ttsk.h")


class TtskError(RuntimeError):
    pass


class GemmDesc(C.Structure):
    """Mirror of `ttsk_gemm_desc` (include/ttsk.h) — field order and types must match exactly."""
    _fields_ = [
        ("A", C.c_void_p), ("B", C.c_void_p), ("C", C.c_void_p), ("C2", C.c_void_p),
        ("bias", C.c_void_p), ("R", C.c_void_p), ("G", C.c_void_p),
        ("M", C.c_int32), ("N", C.c_int32), ("K", C.c_int32),
        ("lda", C.c_int32), ("ldb", C.c_int32), ("ldc", C.c_int32), ("ldr", C.c_int32), ("ldg", C.c_int32),
        ("flags", C.c_int32), ("alpha", C.c_float), ("in_slope", C.c_float), ("out_slope", C.c_float),
        ("nz1", C.c_int32), ("nz2", C.c_int32),
        ("sA1", C.c_int64), ("sA2", C.c_int64), ("sB1", C.c_int64), ("sB2", C.c_int64),
        ("sC1", C.c_int64), ("sC2", C.c_int64), ("sR1", C.c_int64), ("sR2", C.c_int64),
        ("taps", C.c_int32), ("seg_len", C.c_int32), ("tap_shift0", C.c_int32), ("tap_dshift", C.c_int32),
        ("b_tap_stride", C.c_int64),
        ("bseg_len", C.c_int32), ("bshift0", C.c_int32), ("bdshift", C.c_int32),
        ("out_seg", C.c_int32), ("out_mul", C.c_int32), ("out_add", C.c_int32), ("out_add_dz", C.c_int32),
        ("splits", C.c_int32), ("workspace", C.c_void_p), ("workspace_bytes", C.c_int64), ("kernel", C.c_int32),
        ("s_bias1", C.c_int64),
    ]


class ReduceItem(C.Structure):
    """Mirror of `ttsk_reduce_item` (include/ttsk.h)."""
    _fields_ = [("ws", C.c_void_p), ("C", C.c_void_p), ("M", C.c_int32), ("N", C.c_int32), ("ldc", C.c_int32), ("nz", C.c_int32),
                ("splits", C.c_int32), ("accumulate", C.c_int32), ("sC2", C.c_int64), ("alpha", C.c_float)]


class ColsumItem(C.Structure):
    """Mirror of `ttsk_colsum_item` (include/ttsk.h)."""
    _fields_ = [("x", C.c_void_p), ("partials", C.c_void_p), ("is_f32", C.c_int32), ("rows", C.c_int32), ("C", C.c_int32),
                ("ld", C.c_int32), ("nblk", C.c_int32)]


class ScatterItem(C.Structure):
    """Mirror of `ttsk_scatter_item` (include/ttsk.h)."""
    _fields_ = [("dx", C.c_void_p), ("idx", C.c_void_p), ("dtable", C.c_void_p), ("idx_is_i64", C.c_int32), ("idx_div", C.c_int32),
                ("n_idx", C.c_int32), ("n_table_rows", C.c_int32), ("D", C.c_int32), ("skip_row", C.c_int32), ("accumulate", C.c_int32)]


class FinalizeItem(C.Structure):
    """Mirror of `ttsk_finalize_item` (include/ttsk.h)."""
    _fields_ = [("partials", C.c_void_p), ("dst", C.c_void_p), ("nblk", C.c_int32), ("ncols", C.c_int32), ("ld", C.c_int32),
                ("accumulate", C.c_int32), ("scale", C.c_float)]


class PackItem(C.Structure):
    """Mirror of `ttsk_pack_item` (include/ttsk.h)."""
    _fields_ = [("src", C.c_void_p), ("dst", C.c_void_p), ("Cs", C.c_int32), ("K", C.c_int32), ("Ds", C.c_int32), ("transpose", C.c_int32)]


class DwConvItem(C.Structure):
    """Mirror of `ttsk_dwconv_item` (include/ttsk.h)."""
    _fields_ = [("dy", C.c_void_p), ("x", C.c_void_p), ("dw", C.c_void_p), ("lens", C.c_void_p), ("Cout", C.c_int32), ("Cin", C.c_int32),
                ("K", C.c_int32), ("ldy", C.c_int32), ("ldx", C.c_int32), ("B", C.c_int32), ("S", C.c_int32), ("accumulate", C.c_int32)]


class DwGemmItem(C.Structure):
    """Mirror of `ttsk_dwgemm_item` (include/ttsk.h)."""
    _fields_ = [("dy", C.c_void_p), ("x", C.c_void_p), ("dw", C.c_void_p), ("workspace", C.c_void_p), ("lens", C.c_void_p),
                ("Cout", C.c_int32), ("Cin", C.c_int32), ("K", C.c_int32), ("ldy", C.c_int32), ("ldx", C.c_int32), ("B", C.c_int32),
                ("S", C.c_int32), ("accumulate", C.c_int32), ("splits", C.c_int32)]


class AdamItem(C.Structure):
    """Mirror of `ttsk_adam_item` (include/ttsk.h)."""
    _fields_ = [("off", C.c_int64), ("pack", C.c_void_p), ("pack_t", C.c_void_p), ("Cs", C.c_int32), ("K", C.c_int32), ("Ds", C.c_int32),
                ("tile0", C.c_int32)]


# flags (include/ttsk.h)
A_TR, B_TR, C_F32, RELU, ADD_R, R_F32, MASK_G, LRELU_IN, TANH, ACCUM_C, LRELU_OUT, F16, C2_LRELU, DEFER_REDUCE, RAW_SLABS = [1 << i for i in range(15)]


def declared_symbols(header_path=HEADER_PATH):
    """Every function name declared in include/ttsk.h (used by the CPU test that checks the exports)."""
    with open(header_path) as f:
        text = re.sub(r"/\*.*?\*/", "", f.read(), flags=re.S)
    return sorted(set(re.findall(r"\b(ttsk_[a-z0-9_]+)\s*\(", text)))


def _ctype_of(arg):
    arg = arg.strip()
    if arg in ("void", ""):
        return None
    if "*" in arg:
        return C.c_void_p
    for key, ct in (("uint64_t", C.c_uint64), ("int64_t", C.c_int64), ("uint32_t", C.c_uint32), ("int32_t", C.c_int32),
                    ("double", C.c_double), ("float", C.c_float), ("int", C.c_int)):
        if re.search(r"\b%s\b" % key, arg):
            return ct
    raise TtskError("cannot map C parameter %r" % arg)


def declared_prototypes(header_path=HEADER_PATH):
    """name -> ctypes argtypes, parsed from the header so the binding cannot drift from it."""
    with open(header_path) as f:
        text = re.sub(r"/\*.*?\*/", "", f.read(), flags=re.S)
    out = {}
    for m in re.finditer(r"\b(ttsk_[a-z0-9_]+)\s*\(([^;{]*?)\)\s*;", text, flags=re.S):
        args = [a for a in (_ctype_of(x) for x in m.group(2).split(",")) if a is not None]
        out[m.group(1)] = args
    return out


def source_fingerprint():
    """sha256 (first 16 hex digits) over csrc/*.hip, csrc/*.h and include/ttsk.h, by file name then content: the identity of the
    kernel sources a profile was taken with.  tools/pmc_summary.py / pmc_mfma_summary.py store it in profiles/*.json, and bench.py
    flags a committed profile as stale when it differs from the tree it runs in."""
    import glob
    import hashlib
    h = hashlib.sha256()
    csrc = os.path.join(_HERE, "csrc")
    for f in sorted(glob.glob(os.path.join(csrc, "*.hip")) + glob.glob(os.path.join(csrc, "*.h")) + [HEADER_PATH]):
        h.update(os.path.basename(f).encode())
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


_lib = None


def load(path=LIB_PATH):
    """Load the library (once).  Raises TtskError when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(path):
        raise TtskError(
            "libttsk_hip.so not found at %s — build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(or `make -C tts_king_amd/csrc`). There is no CPU fallback for the hot path." % path)
    lib = C.CDLL(path)
    missing = [s for s in declared_symbols() if not hasattr(lib, s)]
    if missing:
        raise TtskError("libttsk_hip.so is stale: missing symbols %s — rebuild" % missing)
    lib.ttsk_last_error.restype = C.c_char_p
    for name, argtypes in declared_prototypes().items():
        fn = getattr(lib, name)
        if name in ("ttsk_resblock_pack_elems", "ttsk_dwgemm_workspace_floats"):
            fn.restype = C.c_int64
        elif name != "ttsk_last_error":
            fn.restype = C.c_int
        fn.argtypes = argtypes
    lib.ttsk_gemm.argtypes = [C.POINTER(GemmDesc), C.c_void_p]
    lib.ttsk_scatter_sum_batch.argtypes = [C.POINTER(ScatterItem), C.c_int, C.c_void_p]
    lib.ttsk_colsum_batch.argtypes = [C.POINTER(ColsumItem), C.c_int, C.c_void_p]
    lib.ttsk_colsum_finalize_batch.argtypes = [C.POINTER(FinalizeItem), C.c_int, C.c_void_p]
    lib.ttsk_gemm_reduce_batch.argtypes = [C.POINTER(ReduceItem), C.c_int, C.c_void_p]
    lib.ttsk_dwconv_batch.argtypes = [C.POINTER(DwConvItem), C.c_int, C.c_void_p]
    lib.ttsk_dwgemm_batch.argtypes = [C.POINTER(DwGemmItem), C.c_int, C.c_int, C.c_void_p]
    lib.ttsk_gemm_plan.argtypes = [C.POINTER(GemmDesc), C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int64)]
    _lib = lib
    return lib


def check(rc, what=""):
    if rc != 0:
        raise TtskError("%s failed (%d): %s" % (what, rc, load().ttsk_last_error().decode()))
