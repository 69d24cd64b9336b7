"""The hot MFMA loops keep the schedule they were given (csrc/tapring.h, FragStream): compiled for gfx950 here (hipcc cross-compiles
without a GPU), no MFMA of the FS2 window conv, the fused LayerNorm kernels or flash attention may sit right behind a wait for an LDS
read issued just before it — what the compiler produces from the plain loop nests (an LDS round trip per MFMA, -4 % of
the train step when it went).  A compiler or source change that brings the pattern back fails here, not in a profile three rounds on."""
import os
import shutil

import pytest

from tools.debug import isa_scan

CSRC = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tts_king_amd", "csrc")
pytestmark = pytest.mark.skipif(not os.path.exists(isa_scan.HIPCC), reason="hipcc not installed")

# file -> (substring of the mangled kernel name, most MFMAs allowed behind `s_waitcnt lgkmcnt(0|1)`, of at least this many MFMAs)
CASES = {
    "ffn_conv.hip": [("win_conv_kernelILi256ELi112ELb0ELb1ELi8ELi2ELb0", 4, 168), ("win_conv_kernelILi512ELi64ELb1ELb1ELi8ELi2ELb0", 4, 96)],
    "flash_attn.hip": [("flash_fwd_t_kernel", 6, 32), ("flash_bwd_t_kernel", 12, 112)],
    "layernorm.hip": [("ln_bwd256_proj_kernelILi4ELb1", 10, 224), ("ln_bwd256_proj_kernelILi1ELb0", 6, 32)],
}


@pytest.mark.parametrize("fname", sorted(CASES))
def test_mfma_operands_are_read_ahead(fname):
    rows = isa_scan.scan(os.path.join(CSRC, fname))
    for sub, most, at_least in CASES[fname]:
        hit = [(k, v) for k, v in rows.items() if sub in k]
        assert hit, "%s: no kernel matching %s among %s" % (fname, sub, sorted(rows))
        for k, (n, behind_lds, behind_vm) in hit:
            assert n >= at_least, (k, n)
            assert behind_lds <= most, "%s: %d of %d MFMAs wait for an LDS read issued just before them (allowed: %d)" % (k, behind_lds, n, most)
            assert behind_vm <= 4, "%s: %d MFMAs behind s_waitcnt vmcnt(0|1): the weight prefetch is gone" % (k, behind_vm)
