"""Where the trainer loop's host time goes (diagnostic): feeder alone, engine.step alone, both."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.getcwd())
from tts_king_amd.config import default_config
from tts_king_amd.dataset import DeviceFeeder
from tts_king_amd.engine import TrainEngine
from tts_king_amd.fastspeech2 import FastSpeech2
from tts_king_amd.loss import FastSpeech2Loss
from tts_king_amd.optimizer import ScheduledOptim
from tts_king_amd.synthetic import make_batch
cfg = default_config(); cfg.train_config["optimizer"]["grad_acc_step"] = 1
dev = "cuda:0"
m = FastSpeech2(cfg.preprocess_config, cfg.model_config, 65, device=dev).train()
opt = ScheduledOptim(m, cfg.train_config, cfg.model_config, 0)
eng = TrainEngine(m, opt, cfg, FastSpeech2Loss(cfg.preprocess_config, cfg.model_config))
host = [tuple(x.numpy() if torch.is_tensor(x) else x for x in make_batch(16, 64 - (i % 3), seed=2000 + i)) for i in range(12)]
bucket = (8, 32, 1000)
t0 = time.perf_counter(); n = 0
for _ in range(5):
    for b in DeviceFeeder(host, dev, bucket=bucket): n += 1
torch.cuda.synchronize(); print("feeder only: %.3f ms/batch" % (1e3 * (time.perf_counter() - t0) / n))
dev_batches = list(DeviceFeeder(host, dev, bucket=bucket))
step = 0
for _ in range(2):
    for b in dev_batches: step += 1; eng.step(b, step)
torch.cuda.synchronize()
t0 = time.perf_counter(); n = 0
for _ in range(5):
    for b in dev_batches: step += 1; n += 1; eng.step(b, step)
torch.cuda.synchronize(); print("engine.step only (device-resident batches): %.3f ms/step" % (1e3 * (time.perf_counter() - t0) / n), eng.stats)
t0 = time.perf_counter(); n = 0
for _ in range(5):
    for b in DeviceFeeder(host, dev, bucket=bucket): step += 1; n += 1; eng.step(b, step)
torch.cuda.synchronize(); print("feeder + engine: %.3f ms/step" % (1e3 * (time.perf_counter() - t0) / n))
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for b in DeviceFeeder(host, dev, bucket=bucket): step += 1; eng.step(b, step)
torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
