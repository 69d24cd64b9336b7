"""Per-shape GEMM time table of one eager train step (diagnostic)."""
import os, sys
import torch
sys.path.insert(0, os.getcwd())
from tts_king_amd import ops
from tts_king_amd.config import default_config
from tts_king_amd.fastspeech2 import FastSpeech2
from tts_king_amd.graph import make_enqueue
from tts_king_amd.loss import FastSpeech2Loss
from tts_king_amd.optimizer import ScheduledOptim
from tts_king_amd.synthetic import make_batch
from tts_king_amd.train_step import to_device
cfg = default_config(); cfg.train_config["optimizer"]["grad_acc_step"] = 1
m = FastSpeech2(cfg.preprocess_config, cfg.model_config, 65, device="cuda:0").train()
opt = ScheduledOptim(m, cfg.train_config, cfg.model_config, 0)
enq = make_enqueue(m, opt, cfg, FastSpeech2Loss(cfg.preprocess_config, cfg.model_config))
b = to_device(make_batch(16, 64, seed=1234), "cuda:0")
for _ in range(3): enq(b)
torch.cuda.synchronize()
tr = []; ops.GEMM_TRACE = tr
for _ in range(3): enq(b)
torch.cuda.synchronize(); ops.GEMM_TRACE = None
agg = {}
for e0, e1, fl, kind, shape in tr:
    d = agg.setdefault((kind,) + shape, [0.0, 0.0, 0]); d[0] += e0.elapsed_time(e1); d[1] += fl; d[2] += 1
print("%-8s %-40s %6s %9s %9s %8s" % ("kind", "M,N,K,taps,batch,splits", "n/step", "us/launch", "ms/step", "TFLOP/s"))
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][0]):
    print("%-8s %-40s %6d %9.1f %9.3f %8.1f" % (k[0], str(k[1:]), v[2] // 3, 1e3 * v[0] / v[2], v[0] / 3, v[1] / (v[0] * 1e-3) / 1e12))
print("total gemm ms/step", sum(v[0] for v in agg.values()) / 3)

# ---- HiFi-GAN generator (B=8, T=384), conv-by-conv and fused
from tts_king_amd.hifi_bench import build_generator
from tts_king_amd.synthetic import make_mel
gen = build_generator(cfg, "cuda:0")
mel = make_mel(8, 384, seed=1234).to("cuda:0")
for fused in (False, True):
    gen.fused = fused
    for _ in range(2): gen(mel)
    torch.cuda.synchronize()
    tr = []; ops.GEMM_TRACE = tr
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3): gen(mel)
    e1.record()
    torch.cuda.synchronize(); ops.GEMM_TRACE = None
    agg = {}
    for a0, a1, fl, kind, shape in tr:
        d = agg.setdefault((kind,) + shape, [0.0, 0.0, 0]); d[0] += a0.elapsed_time(a1); d[1] += fl; d[2] += 1
    print("\nHiFi-GAN fused=%s: eager %.3f ms/run, gemm %.3f ms/run" % (fused, e0.elapsed_time(e1) / 3, sum(v[0] for v in agg.values()) / 3))
    for k, v in sorted(agg.items(), key=lambda kv: -kv[1][0]):
        print("%-8s %-40s %6d %9.1f %9.3f %8.1f" % (k[0], str(k[1:]), v[2] // 3, 1e3 * v[0] / v[2], v[0] / 3, v[1] / (v[0] * 1e-3) / 1e12))
