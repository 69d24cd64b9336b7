"""ms per replayed FS2 step in blocks of 5 replays, from the capture on: how many replays until the time is steady?"""
import os, sys, time, torch
sys.path.insert(0, os.getcwd())
from tts_king_amd.config import default_config
from tts_king_amd.fastspeech2 import FastSpeech2
from tts_king_amd.graph import GraphedTrainStep, make_enqueue
from tts_king_amd.loss import FastSpeech2Loss
from tts_king_amd.optimizer import ScheduledOptim
from tts_king_amd.synthetic import make_batch
from tts_king_amd.train_step import to_device
dev = "cuda:0"; cfg = default_config(); cfg.train_config["optimizer"]["grad_acc_step"] = 1
batch = to_device(make_batch(16, 64, seed=1234), dev)
m = FastSpeech2(cfg.preprocess_config, cfg.model_config, 65, device=dev, seed=1234).train()
o = ScheduledOptim(m, cfg.train_config, cfg.model_config, 0)
g = GraphedTrainStep(make_enqueue(m, o, cfg, FastSpeech2Loss(cfg.preprocess_config, cfg.model_config)), batch, warmup=2)
out = []
for blk in range(12):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5): g.run()
    torch.cuda.synchronize(); out.append(1e3 * (time.perf_counter() - t0) / 5)
print("ms per step, blocks of 5 replays from the capture on:", " ".join("%.3f" % v for v in out))
