"""Diagnostic: what the per-step input copies cost a replayed train step — the hipGraph replay alone, with the batch copied into the
graph's static inputs tensor by tensor (TrainEngine's path), and with one flat copy.  usage: python tools/debug/replay_copy_cost.py"""
import os, sys, time
import torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.getcwd()))
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
batch = to_device(make_batch(16, 64, seed=1234), DEV)
g = GraphedTrainStep(make_enqueue(m, opt, cfg, FastSpeech2Loss(cfg.preprocess_config, cfg.model_config)), batch, warmup=2)
other = [t.clone() if torch.is_tensor(t) else t for t in batch]
n_t = sum(1 for t in other if torch.is_tensor(t))

def timed(fn, n=200):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / n
print("replay only:                      %.4f ms" % timed(lambda: g.run()))
print("replay + %d tensor copies (D2D):  %.4f ms" % (n_t, timed(lambda: g.run(other))))
