"""HiFi-GAN generator B=8, T=384 under graph replay: ms per batch (the bench.py `hifigan` leg alone)."""
import os, sys, json
import torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.getcwd()))
from tts_king_amd.config import default_config
from tts_king_amd.hifi_bench import hifi_rtf
r = hifi_rtf(default_config(), "cuda:0", iters=30)
print(json.dumps({k: r[k] for k in ("ms_per_batch", "device_ms_per_batch", "rtf", "tflops")}))
