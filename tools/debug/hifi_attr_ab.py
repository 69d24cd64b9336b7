"""Generator attributes A/B'd on the replayed HiFi-GAN batch (B = 8, 384 mel frames), alternated on one box:
python tools/debug/hifi_attr_ab.py name=value[,name=value] ...   e.g.  loop_upsample=False   window_conv_pre=False"""
import os, sys, time
import torch
sys.path.insert(0, os.getcwd())
from tts_king_amd.config import default_config
from tts_king_amd.hifi_bench import build_generator
DEV = "cuda:0"
cfg = default_config()
mel = (torch.randn(8, 80, 384, device=DEV) * 2.0 - 5.0) if os.environ.get("MEL_RANGE", "1") == "1" else torch.randn(8, 80, 384, device=DEV)      # bench.py's input statistics (synthetic.make_mel) by default


def build(attrs):
    gen = build_generator(cfg, DEV)
    for kv in attrs:
        k, v = kv.split("=")
        assert hasattr(gen, k), k
        setattr(gen, k, eval(v))
    gen._packed = None
    gen(mel)
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        y = gen(mel)
    return gr, gen, y


def t(g, n=50):
    for _ in range(5):
        g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        g.replay()
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / n


variants = {"default": ()}
for arg in sys.argv[1:]:
    variants[arg] = tuple(arg.split(","))
N = int(os.environ.get("INSTANCES", "1"))          # captured graphs per variant (two instances of one variant differ by several us)
graphs = {k: [build(v) for _ in range(N)] for k, v in variants.items()}
for r in range(4):
    out = []
    for k, gs in graphs.items():
        ts = [t(g[0]) for g in gs]
        out.append("%s %.4f ms" % (k, sum(ts) / len(ts)) + ("" if N == 1 else " (%s)" % " ".join("%.4f" % x for x in ts)))
    print(" | ".join(out), flush=True)
