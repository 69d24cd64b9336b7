"""A/B on one box: HiFi-GAN ms per batch (graph replay) with the fused last stage (csrc/mrf32.hip) on / off, and the per-stage device
times of both."""
import os, sys
import torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.getcwd()))
from tts_king_amd.config import default_config
from tts_king_amd.hifi_bench import build_generator, stage_rooflines
from tts_king_amd.synthetic import make_mel

cfg, dev = default_config(), "cuda:0"
gen = build_generator(cfg, dev)
mel = make_mel(8, 384, seed=1234).to(dev)


def timed(iters=30):
    for _ in range(3):
        gen(mel)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        gen(mel)
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        g.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


for rnd in range(2):
    for fused in (True, False):
        gen.mrf_fused = fused
        ms = timed()
        st = stage_rooflines(gen, mel, 8, 384, iters=3)
        print("mrf_fused=%s: %.3f ms per batch; stages: %s" % (fused, ms, {k: round(v["ms"], 3) for k, v in st.items()}))
