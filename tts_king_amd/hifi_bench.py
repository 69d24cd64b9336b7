"""HiFi-GAN generator timing for bench.py (BASELINE.json configs[2]: B=8, 80-bin mel x 384 frames -> 8 x 98,304
samples at 22.05 kHz).  RTF = wall seconds / audio seconds with the mel resident in HBM; the int16 D2H copy of
`HIFIapi.generate` is reported separately."""
import time

import torch

from .hifigan import Generator
from .synthetic import make_mel

HIFI_FLOP_PER_FRAME = 614.11e6          # SURVEY.md §8d (verified with torch.utils.flop_counter on the reference)


def build_generator(cfg, dev, seed=1234):
    g = Generator(cfg.hifi)
    g.reset_parameters(seed)
    g.to(dev)
    g.remove_weight_norm()
    return g.eval()


def hifi_rtf(cfg, dev, B=8, T=384, iters=20, warmup=3, use_graph=True):
    gen = build_generator(cfg, dev)
    mel = make_mel(B, T, seed=1234).to(dev)
    for _ in range(warmup):
        wav = gen(mel)
    torch.cuda.synchronize()
    run = lambda: gen(mel)
    if use_graph:
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            wav = gen(mel)
        run = graph.replay
        run()
        torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    for _ in range(iters):
        run()
    e1.record()
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / iters
    dev_s = e0.elapsed_time(e1) * 1e-3 / iters
    sr = float(cfg.hifi.sampling_rate)
    samples = B * T * 256
    audio_s = samples / sr
    # PCIe-inclusive variant: device int16 conversion + D2H, as HIFIapi.generate does
    from . import ops
    t1 = time.perf_counter()
    i16 = ops.to_int16(wav, float(cfg.hifi.MAX_WAV_VALUE)).cpu()
    d2h = time.perf_counter() - t1
    flops = HIFI_FLOP_PER_FRAME * B * T
    return {"workload": "HiFi-GAN V1 generator, B=%d, T=%d mel frames -> %d samples @ %d Hz (BASELINE.json configs[2])" % (B, T, samples, int(sr)),
            "rtf": wall / audio_s, "ms_per_batch": 1e3 * wall, "device_ms_per_batch": 1e3 * dev_s, "samples_per_s": samples / wall,
            "audio_seconds_per_batch": audio_s, "tflops": flops / wall / 1e12, "mfma_roofline_frac": flops / wall / 2.5e15,
            "int16_d2h_ms": 1e3 * d2h, "launch": "hipGraph replay" if use_graph else "eager", "dtype": "f16 (fp32 accumulate)",
            "iters": iters}
