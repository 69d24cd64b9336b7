"""Host-side cost of the eagerly launched FS2 step: cProfile over 30 steps (which Python frames the ~150 launches go through)."""
import cProfile
import pstats
import sys
import time

import torch

sys.path.insert(0, ".")
from tts_king_amd.config import default_config
from tts_king_amd.fastspeech2 import FastSpeech2
from tts_king_amd.graph import make_enqueue
from tts_king_amd.loss import FastSpeech2Loss
from tts_king_amd.optimizer import ScheduledOptim
from tts_king_amd.synthetic import make_batch
from tts_king_amd.train_step import to_device

cfg = default_config()
cfg.train_config["optimizer"]["grad_acc_step"] = 1
dev = "cuda:0"
model = FastSpeech2(cfg.preprocess_config, cfg.model_config, 65, device=dev, seed=1234).train()
opt = ScheduledOptim(model, cfg.train_config, cfg.model_config, 0)
enq = make_enqueue(model, opt, cfg, FastSpeech2Loss(cfg.preprocess_config, cfg.model_config))
batch = to_device(make_batch(16, 64, seed=1234), dev)
for _ in range(5):
    enq(batch)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(30):
    enq(batch)
torch.cuda.synchronize()
print("eager ms/step", 1e3 * (time.perf_counter() - t0) / 30)
pr = cProfile.Profile()
pr.enable()
for _ in range(30):
    enq(batch)
torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(22)
