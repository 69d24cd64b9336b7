"""ms per replayed HiFi-GAN batch (B = 8, 384 frames) in blocks of 10 replays, from the capture on."""
import os, sys, time, torch
sys.path.insert(0, os.getcwd())
from tts_king_amd.config import default_config
from tts_king_amd.hifi_bench import build_generator
from tts_king_amd.synthetic import make_mel
cfg = default_config(); dev = "cuda:0"
gen = build_generator(cfg, dev); mel = make_mel(8, 384, seed=1234).to(dev)
for _ in range(3): gen(mel)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g): wav = gen(mel)
out = []
for blk in range(12):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): g.replay()
    torch.cuda.synchronize(); out.append(1e3 * (time.perf_counter() - t0) / 10)
print("ms per batch, blocks of 10 replays from the capture on:", " ".join("%.3f" % v for v in out))
