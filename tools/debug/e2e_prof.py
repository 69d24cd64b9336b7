"""The e2e synthesis path of bench.py (one utterance, replayed graphs) for rocprofv3 (diagnostic)."""
import os, sys
import torch
R = os.environ.get("GRAFT_REPO_ROOT", os.getcwd())
sys.path.insert(0, R)
import bench
from tts_king_amd.config import default_config
cfg = default_config()
print(bench.e2e_synth_leg(cfg, "cuda:0", iters=10, with_cpu=False))
