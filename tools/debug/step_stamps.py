"""Diagnostic: the phases of a REPLAYED train step on the device's own clock, without the profiler (whose presence changed the
step's period by 0.25 ms once the step ran three branches): one-thread stamp kernels at the marks ops.stamp() sets.
Needs `make -C tts_king_amd/csrc stamps` and TTSK_LIB_PATH=tts_king_amd/libttsk_hip_stamps.so.  usage: python tools/debug/step_stamps.py"""
import os, sys, copy
import torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.getcwd()))
from tts_king_amd import ops
from tts_king_amd.config import default_config
from tts_king_amd.fastspeech2 import FastSpeech2
from tts_king_amd.graph import GraphedTrainStep, make_enqueue
from tts_king_amd.loss import FastSpeech2Loss
from tts_king_amd.optimizer import ScheduledOptim
from tts_king_amd.synthetic import make_batch
from tts_king_amd.train_step import to_device
DEV = "cuda:0"
cfg = default_config(); cfg.train_config["optimizer"]["grad_acc_step"] = 1
m = FastSpeech2(cfg.preprocess_config, cfg.model_config, 65, device=DEV, seed=1234).train()
opt = ScheduledOptim(m, cfg.train_config, cfg.model_config, 0)
loss = FastSpeech2Loss(cfg.preprocess_config, cfg.model_config)
batch = to_device(make_batch(16, 64, seed=1234), DEV)
enq = make_enqueue(m, opt, cfg, loss)
for _ in range(2):
    enq(batch)
torch.cuda.synchronize()
buf = torch.zeros(64, dtype=torch.int64, device=DEV)
names = []
ops.STAMPS = (buf, names)
g = GraphedTrainStep(enq, batch, warmup=0)
ops.STAMPS = None
for _ in range(10):
    g.run()
torch.cuda.synchronize()
acc = {}
N = 20
for _ in range(N):
    g.run()
    torch.cuda.synchronize()
    v = buf.cpu().tolist()
    t0 = v[0]
    for n, t in zip(names, v):
        acc[n] = acc.get(n, 0.0) + (t - t0) / 100.0
import time
t0 = time.perf_counter()
for _ in range(50):
    g.run()
torch.cuda.synchronize()
print("replay period (50 back-to-back, with the stamp kernels in): %.1f us" % (1e6 * (time.perf_counter() - t0) / 50))
for n in names:
    print("%-22s %9.1f us" % (n, acc[n] / N))
