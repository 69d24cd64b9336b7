"""A/B on one box: HiFi-GAN ms per batch (graph replay) with the pair kernel disabled per channel count (TTSK_AB_NO_PAIR=256,64,...)."""
import os, sys, json
import torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.getcwd()))
from tts_king_amd import ops
from tts_king_amd.config import default_config
from tts_king_amd.hifi_bench import hifi_rtf
orig = ops.hifi_conv_pair_supported
for off in ([], [32], [], [32], [256, 128, 64, 32]):
    ops.hifi_conv_pair_supported = (lambda C, K, d, off=off: (C not in off) and orig(C, K, d))
    r = hifi_rtf(default_config(), "cuda:0", iters=30)
    print("pair kernel off for C in %-16s: %.3f ms per batch" % (off, r["ms_per_batch"]))
