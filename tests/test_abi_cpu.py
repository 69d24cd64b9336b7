"""CPU: the C-ABI shared library loads and exports every symbol include/ttsk.h declares (no compute)."""
import ctypes

from tts_king_amd import lib


def test_exports_match_header():
    names = lib.declared_symbols()
    assert "ttsk_gemm" in names and "ttsk_length_regulator_fwd" in names
    l = lib.load()
    for n in names:
        assert hasattr(l, n), n
    assert l.ttsk_version() == 1
    assert set(lib.declared_prototypes()) == set(names)


def test_gemm_desc_layout_matches_header():
    """sizeof(ttsk_gemm_desc) as laid out by ctypes must equal the C compiler's (checked through a null call)."""
    d = lib.GemmDesc()
    assert ctypes.sizeof(d) % 8 == 0
    rc = lib.load().ttsk_gemm(ctypes.byref(d), None)       # all-zero descriptor: rejected, nothing launched
    assert rc == -1
    assert b"null operand" in lib.load().ttsk_last_error()


def test_missing_library_fails_loudly(tmp_path):
    import pytest
    saved = lib._lib
    lib._lib = None
    try:
        with pytest.raises(lib.TtskError):
            lib.load(str(tmp_path / "nope.so"))
    finally:
        lib._lib = saved
