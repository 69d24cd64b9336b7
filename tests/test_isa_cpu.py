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
    # (round 6: ...ELb1ELi2 = the SPLIT instance, two output groups per workgroup: the upstream projection's 96 MFMAs + two groups' 64)
    "layernorm.hip": [("ln_bwd256_proj_kernelILi4ELb1ELi1", 10, 224), ("ln_bwd256_proj_kernelILi4ELb1ELi2", 8, 160), ("ln_bwd256_proj_kernelILi1ELb0", 6, 32)],
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


# ---- registers decide the workgroups per CU, and those the ROUNDS of a launch: the kernels whose design rests on a count
# file -> (substring of the mangled kernel name, most VGPRs, most spilled VGPRs)
from tools.debug import occupancy      # noqa: E402

REGISTER_BOUNDS = {
    # 848 workgroups at the training shape need four per CU to run in one round (130-136 registers gave three: a second round for 80)
    "batchnorm.hip": [("bn_apply2_kernelILb0", 128, 0), ("bn_bwd_apply2_kernel", 128, 4)],
    # two 4-wave workgroups per CU (the compiler took 316 when left alone); the looped upsampler holds four weight sets without spilling
    "ffn_conv.hip": [("win_conv_kernelILi128ELi224ELb0ELb1ELi4ELi2ELb1", 256, 0), ("ups_loop_kernelILi256ELi96", 256, 0)],
    # the predictors' LayerNorm backward beside the PostNet's kernels: two 8-wave workgroups per CU
    "layernorm.hip": [("ln_bwd_kernelILi1", 128, 0)],
    # two pair workgroups per CU (convwin.hip: one's window load under the other's taps)
    "convwin.hip": [("conv_pair_kernelILb1", 256, 0), ("conv_pair_fs_kernelILi64ELi4ELb1", 256, 0)],
}


@pytest.mark.parametrize("fname", sorted(REGISTER_BOUNDS))
def test_register_counts_keep_the_designed_workgroups_per_cu(fname):
    rows = {name: (vg, sp) for name, lds, wg, vg, sp, by_v, by_l in occupancy.scan(os.path.join(CSRC, fname))}
    for sub, most, most_spilled in REGISTER_BOUNDS[fname]:
        hit = [(k, v) for k, v in rows.items() if sub in k]
        assert hit, "%s: no kernel matching %s among %s" % (fname, sub, sorted(rows))
        for k, (vg, sp) in hit:
            assert vg <= most, "%s: %d registers (the design needs <= %d)" % (k, vg, most)
            assert sp <= most_spilled, "%s: %d spilled registers (allowed: %d)" % (k, sp, most_spilled)
