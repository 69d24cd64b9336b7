"""The trainer loop (TrainEngine over DeviceFeeder, as bench.py's `train_loop` leg) with the host time per step spent in the feeder's
__next__ (staging batch k + 1) and in TrainEngine.step (copy + replay launch), beside the wall time per step."""
import copy, os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.getcwd()))
from tts_king_amd.config import default_config
from tts_king_amd.dataset import DeviceFeeder
from tts_king_amd.engine import TrainEngine
from tts_king_amd.fastspeech2 import FastSpeech2
from tts_king_amd.loss import FastSpeech2Loss
from tts_king_amd.optimizer import ScheduledOptim
from tts_king_amd.synthetic import make_batch

cfg, dev = default_config(), "cuda:0"
c1 = copy.deepcopy(cfg); c1.train_config["optimizer"]["grad_acc_step"] = 1
tb = int(os.environ.get("T_BUCKET", "32"))
m = FastSpeech2(c1.preprocess_config, c1.model_config, 65, device=dev, seed=1234).train()
o = ScheduledOptim(m, c1.train_config, c1.model_config, 0)
eng = TrainEngine(m, o, c1, FastSpeech2Loss(c1.preprocess_config, c1.model_config))
bucket = (8, tb, int(cfg.model_config["max_seq_len"]))
host = [tuple(x.numpy() if torch.is_tensor(x) else x for x in make_batch(16, 64 - (i % 3), seed=2000 + i)) for i in range(12)]
step = [0]
tn, ts = [0.0], [0.0]

def run(batches, timed=False):
    it = iter(DeviceFeeder(batches, dev, bucket=bucket))
    last = None
    while True:
        t0 = time.perf_counter()
        try:
            b = next(it)
        except StopIteration:
            break
        t1 = time.perf_counter()
        step[0] += 1
        last, _ = eng.step(b, step[0])
        t2 = time.perf_counter()
        if timed:
            tn[0] += t1 - t0; ts[0] += t2 - t1
    return last

run(host); run(host); torch.cuda.synchronize()
n_loops = 8
t0 = time.perf_counter()
for _ in range(n_loops):
    last = run(host, True)
last.cpu()
dt = time.perf_counter() - t0
n = n_loops * len(host)
print("t_bucket %d: %.3f ms per step wall; host in feeder.__next__ %.3f ms, in engine.step %.3f ms; graphs %d; %s" % (
    tb, 1e3 * dt / n, 1e3 * tn[0] / n, 1e3 * ts[0] / n, len(eng._graphs), dict(eng.stats)))
if os.environ.get("LOOP_PROFILE"):
    import cProfile, pstats
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(4):
        last = run(host)
    last.cpu()
    pr.disable()
    pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
if os.environ.get("LOOP_EVENTS"):
    import tts_king_amd.dataset as D
    orig = D._PinnedPool.__init__
    for rep in range(2):
        for depth in (3, 6):
            D._POOLS.clear()
            D._DEBUG_EVENTS = ev = []
            D._PinnedPool.__init__ = lambda self, depth=depth: orig(self, depth)
            run(host); torch.cuda.synchronize()
            del ev[:]
            t0 = time.perf_counter()
            for _ in range(6):
                last = run(host)
            last.cpu()
            dt = time.perf_counter() - t0
            done = sum(1 for q, _ in ev if q)
            print("depth %d: %.3f ms per step; %d slot waits, %d complete at query; synchronize: mean %.3f ms, max %.3f ms" % (
                depth, 1e3 * dt / (6 * len(host)), len(ev), done, 1e3 * sum(t for _, t in ev) / max(len(ev), 1), 1e3 * max([t for _, t in ev] or [0])))
    D._PinnedPool.__init__ = orig
