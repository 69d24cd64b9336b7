"""Every environment switch of the product path, in ONE table (name -> default, meaning).  Nothing under csrc/ reads the
environment: kernels take what they need as arguments.  Each switch selects between two tested paths; the defaults are the measured
optimum.  Paths that two rounds of A/B retired were deleted with their switches in round 4 (the pre-flash attention
kernels, the round-2 flash kernels, the "early" data-parallel schedule, the side-stream placement variants)."""
import os

TABLE = {
    # name: (default, meaning)
    "TTSK_LIB_PATH": ("", "another build of libttsk_hip.so (diagnostic builds of tools/debug: `make stamps`, `make asan`); empty = the in-tree library"),
    "TTSK_WINDOW_FFN": ("1", "FS2: window-kernel convs / projections and their weight packs; 0 = the implicit-GEMM kernels everywhere"),
    "TTSK_DWCONV": ("1", "FS2: weight gradients on csrc/dwconv.hip (w_1) and csrc/dwgemm.hip (other 256-multiple shapes); 0 = grouped GEMMs"),
    "TTSK_DW_SIDE_WGS": ("192", "FS2: grid cap of the weight-gradient work on the second stream beside the encoder-side chain; 0 = no second stream"),
    "TTSK_DW_SIDE_FRAC": ("1.0", "FS2: share of the queued grouped-GEMM FLOPs launched on the second stream (the rest joins the final phase)"),
    "TTSK_PRED_SIDE": ("1", "FS2: the predictors' forward / backward on a stream of their own: 1 both, f forward only, b backward only, 0 neither"),
    "TTSK_DP_SCHEDULE": ("side", "data parallel: side = buckets announced from the second stream behind the decoder-side weight gradients; "
                                 "late = every all-reduce after the last weight-gradient launch"),
    "TTSK_DIST_TIMEOUT_S": ("300", "data parallel: bound on every collective of the job in seconds (parallel.init_distributed): a rank that waits "
                                   "longer fails, the job exits non-zero"),
    "TTSK_CTRL_TIMEOUT_S": ("21600", "data parallel: bound on control-plane waits in seconds (parallel.ControlPlane: validation sums, the barrier behind rank 0's "
                                     "checkpoint write); separate from, and much larger than, TTSK_DIST_TIMEOUT_S"),
    "TTSK_TEST_CHILD": ("", "tests only: set in the child pytest process of tests/conftest.py: isolated"),
    "TTSK_DP_GRAPH": ("1", "bench.py --gpus N: capture the data-parallel step (RCCL all-reduces included) in a hipGraph; 0 = eager launches"),
    "TTSK_HIFI_UPS8": ("1", "HiFi-GAN: stride-8 upsamplers and 128 -> 64 on the window-conv kernel; 0 = polyphase GEMMs / streaming kernel"),
    "TTSK_CPU_THREADS": ("16", "bench.py: threads of the CPU baseline leg (capped at the host's cores)"),
}


def get(name):
    """Value of switch `name` (environment, else the table's default).  Unknown names are a programming error."""
    return os.environ.get(name, TABLE[name][0])


def unknown_in_environment():
    """TTSK_* variables set in the environment that no code reads (typos, switches of deleted paths): reported by bench.py."""
    return sorted(k for k in os.environ if k.startswith("TTSK_") and k not in TABLE)
