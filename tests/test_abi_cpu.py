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


ASAN_RT = "/opt/rocm/lib/llvm/lib/clang/22/lib/linux/libclang_rt.asan-x86_64.so"


def _asan_run(extra=()):
    import glob
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    so = os.path.join(root, "tts_king_amd", "libttsk_hip_asan.so")
    if os.path.exists("/opt/rocm/bin/hipcc"):       # bring the ASan build up to date with the sources (a no-op when it is)
        subprocess.run(["make", "-C", os.path.join(root, "tts_king_amd", "csrc"), "-j", "8", "asan"], stdout=subprocess.DEVNULL,
                       stderr=subprocess.DEVNULL, timeout=1500)
    rts = [ASAN_RT] if os.path.exists(ASAN_RT) else glob.glob("/opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so")
    if not os.path.exists(so) or not rts:
        import pytest
        pytest.skip("no ASan host build (make -C tts_king_amd/csrc asan; __graft_entry__.build() makes it)")
    env = dict(os.environ, LD_PRELOAD=rts[0], ASAN_OPTIONS="detect_leaks=0:verify_asan_link_order=0", TTSK_LIB_PATH=so)
    return subprocess.run([sys.executable, os.path.join(root, "tests", "abi_sweep.py")] + list(extra), env=env,
                          stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)


def test_argument_checking_under_address_sanitizer():
    """the host side of the C-ABI library built with -fsanitize=address (CPU build only; GPU ASan is not available
    on this pool).  tests/abi_sweep.py calls every declared entry point with null / zero arguments (each must refuse before its
    launch), sweeps the host-only size queries, and runs the planner and the grouped-launch table builder into exactly sized
    host buffers."""
    p = _asan_run()
    assert p.returncode == 0, p.stderr.decode(errors="replace")[-3000:]
    assert b"abi sweep ok" in p.stdout and b"AddressSanitizer" not in p.stderr


def test_address_sanitizer_is_live():
    """Negative control: the same sweep with a table buffer 64 bytes too small is reported (heap-buffer-overflow) and aborts."""
    p = _asan_run(["--overrun"])
    assert p.returncode != 0 and b"AddressSanitizer" in p.stderr and b"overrun not detected" not in p.stdout
