"""Soak run of the trainer's own loop (TrainEngine + DeviceFeeder, hipGraph replay per shape bucket, dropout on): N passes over 12
synthetic batches of varying shape; prints the losses every 20 passes; fails on a non-finite value or if the total loss has not
fallen (12 batches are memorised quickly).  usage: python tools/debug/soak.py [passes]"""
import copy
import math
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.getcwd())
from tts_king_amd.config import default_config
from tts_king_amd.dataset import DeviceFeeder
from tts_king_amd.engine import TrainEngine
from tts_king_amd.fastspeech2 import FastSpeech2
from tts_king_amd.loss import FastSpeech2Loss
from tts_king_amd.optimizer import ScheduledOptim
from tts_king_amd.synthetic import make_batch

passes = int(sys.argv[1]) if len(sys.argv) > 1 else 120
dev = "cuda:0"
cfg = default_config()
cfg.train_config["optimizer"]["grad_acc_step"] = 1
model = FastSpeech2(cfg.preprocess_config, cfg.model_config, 65, device=dev, seed=1234).train()
opt = ScheduledOptim(model, cfg.train_config, cfg.model_config, 0)
eng = TrainEngine(model, opt, cfg, FastSpeech2Loss(cfg.preprocess_config, cfg.model_config))
mi = cfg.get("mi355x", {})
bucket = (int(mi.get("l_bucket", 8)), int(mi.get("t_bucket", 32)), int(cfg.model_config["max_seq_len"]))
host = [tuple(x.numpy() if torch.is_tensor(x) else x for x in make_batch(16, 64 - (i % 3), seed=3000 + i)) for i in range(12)]
step, first, t0 = 0, None, time.perf_counter()
for p in range(passes):
    tot = np.zeros(5)
    for b in DeviceFeeder(host, dev, bucket=bucket):
        step += 1
        losses, _ = eng.step(b, step)
        tot += np.asarray(losses[:5].cpu().tolist())
    tot /= len(host)
    if not all(math.isfinite(v) for v in tot):
        raise SystemExit("non-finite loss in pass %d: %s" % (p, tot))
    if first is None:
        first = tot.copy()
    if p % 20 == 0 or p == passes - 1:
        print("pass %4d step %5d  total %.4f  mel %.4f  pitch %.4f  energy %.4f  duration %.4f  lr %.2e  |g| %.3f" % (p, step, *tot, opt.lr(), opt.grad_norm()), flush=True)
torch.cuda.synchronize()
print("%d steps in %.1f s (%.2f ms/step incl. the per-pass host reads), engine %s" % (step, time.perf_counter() - t0, 1e3 * (time.perf_counter() - t0) / step, dict(eng.stats)))
assert tot[0] < 0.6 * first[0], ("total loss did not fall", first, tot)
flat = model.flat_buffers()[0]
assert bool(torch.isfinite(flat).all()), "non-finite parameter"
print("soak ok")
