"""Model attributes A/B'd on the replayed FS2 step, alternated on one box: python tools/debug/attr_ab.py name=value[;name=value] ...
(each argument one variant beside the default), e.g.  split_loss=False   'p_enc=0;p_dec=0'   'ops.DWG_TARGET_STEPS=[112]*2'"""
import copy, os, sys, time
import torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.getcwd()))
from tts_king_amd.config import default_config
from tts_king_amd.fastspeech2 import FastSpeech2
from tts_king_amd.graph import GraphedTrainStep, make_enqueue
from tts_king_amd.loss import FastSpeech2Loss
from tts_king_amd.optimizer import ScheduledOptim
from tts_king_amd.synthetic import make_batch
from tts_king_amd.train_step import to_device

dev = "cuda:0"
cfg = default_config()
cfg.train_config["optimizer"]["grad_acc_step"] = 1
batch = to_device(make_batch(16, 64, seed=1234), dev)


def build(attrs):
    m = FastSpeech2(cfg.preprocess_config, cfg.model_config, 65, device=dev, seed=1234).train()
    for kv in attrs:
        k, v = kv.split("=")
        if k.startswith("ops."):          # a module-level knob of tts_king_amd.ops, in force while THIS variant's graph is captured
            from tts_king_amd import ops as _ops
            setattr(_ops, k[4:], eval(v))
        else:
            setattr(m, k, eval(v))
    if attrs:
        m._build_packs()                 # (attributes that decide which weight packs exist)
        m.sync_shadow(force=True)
    o = ScheduledOptim(m, cfg.train_config, cfg.model_config, 0)
    enq = make_enqueue(m, o, cfg, FastSpeech2Loss(cfg.preprocess_config, cfg.model_config))
    g = GraphedTrainStep(enq, batch, warmup=2)
    g.keepalive = (m, o, enq)          # the graph's kernels point into the model's and the optimizer's buffers
    for kv in attrs:                   # module-level knobs back to their defaults for the next variant
        if kv.startswith("ops."):
            from tts_king_amd import ops as _ops
            setattr(_ops, kv.split("=")[0][4:], DEFAULTS[kv.split("=")[0][4:]])
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


from tts_king_amd import ops as _ops0
DEFAULTS = {k: getattr(_ops0, k) for k in dir(_ops0) if k.isupper()}
variants = {"default": ()}
for arg in sys.argv[1:]:
    variants[arg] = tuple(arg.split(";"))          # (";" between the settings of one variant: values may hold commas)
# INSTANCES > 1: that many captured graphs per variant (each capture lands its buffers and its queues differently: +-15 us between two
# instances of the SAME variant), their mean and spread reported
N = int(os.environ.get("INSTANCES", "1"))
graphs = {k: [build(v) for _ in range(N)] for k, v in variants.items()}
for r in range(3):
    out = []
    for k, gs in graphs.items():
        ts = [t(g) for g in gs]
        out.append("%s %.4f ms" % (k, sum(ts) / len(ts)) + ("" if N == 1 else " (%s)" % " ".join("%.4f" % x for x in ts)))
    print(" | ".join(out), flush=True)
