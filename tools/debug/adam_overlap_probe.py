"""What would a deferred optimizer slice cost if it ran beside the encoder forward of the next step?  The replayed FS2 step with an
Adam-shaped HBM stream over N floats (tools/debug/stream_probe.hip: read p, g, m, v, write p, m, v) launched at the step's start
 - on a second stream (joined at the end of the step): the interference with the latency-bound encoder chain;
 - on the step's own stream: the stream's own duration in the graph.
python tools/debug/adam_overlap_probe.py [grid ...]   (default grids: 2048 256 96)"""
import ctypes, os, sys, time
import torch
root = os.environ.get("GRAFT_REPO_ROOT", os.getcwd())
sys.path.insert(0, root)
from tts_king_amd.config import default_config
from tts_king_amd.fastspeech2 import FastSpeech2
from tts_king_amd.graph import GraphedTrainStep, make_enqueue
from tts_king_amd.loss import FastSpeech2Loss
from tts_king_amd.optimizer import ScheduledOptim
from tts_king_amd.synthetic import make_batch
from tts_king_amd.train_step import to_device

so = os.path.join(root, "tools/debug/stream_probe.so")
if not os.path.exists(so):          # (built artefacts are not in the history: build it where hipcc is — the container, before gpurun ships the tree)
    import subprocess
    subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", so, os.path.join(root, "tools/debug/stream_probe.hip")], check=True)
lib = ctypes.CDLL(so)
lib.probe_launch.argtypes = [ctypes.c_void_p] * 4 + [ctypes.c_longlong, ctypes.c_int, ctypes.c_void_p]
dev = "cuda:0"
cfg = default_config()
cfg.train_config["optimizer"]["grad_acc_step"] = 1
batch = to_device(make_batch(16, 64, seed=1234), dev)
N = int(os.environ.get("PROBE_FLOATS", str(21_400_000))) // 4 * 4
bufs = [torch.rand(N, device=dev) for _ in range(4)]
side = torch.cuda.Stream(device=dev)


def probe(grid):
    rc = lib.probe_launch(bufs[0].data_ptr(), bufs[1].data_ptr(), bufs[2].data_ptr(), bufs[3].data_ptr(), N, grid,
                          torch.cuda.current_stream().cuda_stream)
    assert rc == 0, rc


def build(mode, grid):
    m = FastSpeech2(cfg.preprocess_config, cfg.model_config, 65, device=dev, seed=1234).train()
    o = ScheduledOptim(m, cfg.train_config, cfg.model_config, 0)
    enq = make_enqueue(m, o, cfg, FastSpeech2Loss(cfg.preprocess_config, cfg.model_config))

    def wrapped(b):
        cur = torch.cuda.current_stream()
        if mode == "side":
            side.wait_stream(cur)
            with torch.cuda.stream(side):
                probe(grid)
        elif mode == "serial":
            probe(grid)
        r = enq(b)
        if mode == "side":
            cur.wait_stream(side)
        return r
    g = GraphedTrainStep(wrapped, batch, warmup=2)
    g.keepalive = (m, o, enq, wrapped)
    return g


def t(g, n=200):
    for _ in range(20):
        g.run()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        g.run()
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / n


grids = [int(a) for a in sys.argv[1:]] or [2048, 256, 96]
graphs = {"default": build("none", 0)}
for G in grids:
    graphs["serial/%d" % G] = build("serial", G)
    graphs["side/%d" % G] = build("side", G)
for r in range(3):
    ts = {k: t(g) for k, g in graphs.items()}
    base = ts["default"]
    print("default %.4f ms | " % base + " | ".join("%s %+.1f us" % (k, 1e3 * (v - base)) for k, v in ts.items() if k != "default"), flush=True)
