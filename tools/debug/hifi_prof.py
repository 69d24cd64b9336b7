"""HiFi-GAN generator, B=8 T=384, a few runs (for rocprofv3)."""
import os, sys
import torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.getcwd()))
from tts_king_amd.config import default_config
from tts_king_amd.hifi_bench import build_generator
from tts_king_amd.synthetic import make_mel
cfg = default_config()
gen = build_generator(cfg, "cuda:0")
mel = make_mel(8, 384, seed=1234).to("cuda:0")
for _ in range(6):
    gen(mel)
torch.cuda.synchronize()
