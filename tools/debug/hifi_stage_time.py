"""HiFi-GAN ms per batch (graph replay) and per-stage device times under the current environment switches (one line)."""
import os, sys
import torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.getcwd()))
from tts_king_amd.config import default_config
from tts_king_amd.hifi_bench import build_generator, stage_rooflines
from tts_king_amd.synthetic import make_mel

cfg, dev = default_config(), "cuda:0"
gen = build_generator(cfg, dev)
mel = make_mel(8, 384, seed=1234).to(dev)
for _ in range(3):
    gen(mel)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    gen(mel)
g.replay(); torch.cuda.synchronize()
best = 1e9
for rep in range(3):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(30):
        g.replay()
    e1.record(); torch.cuda.synchronize()
    best = min(best, e0.elapsed_time(e1) / 30)
st = stage_rooflines(gen, mel, 8, 384, iters=3)
print("%s: %.3f ms per batch; stages: %s" % (" ".join("%s=%s" % (k, v) for k, v in os.environ.items() if k.startswith("TTSK_")), best,
                                             {k: round(v["ms"], 3) for k, v in st.items()}))
