"""HiFi-GAN generator timing for bench.py (BASELINE.json configs[2]: B=8, 80-bin mel x 384 frames -> 8 x 98,304
samples at 22.05 kHz).  RTF = wall seconds / audio seconds with the mel resident in HBM; the int16 D2H copy of
`HIFIapi.generate` is reported separately."""
import time

import torch

from .hifigan import Generator
from .synthetic import make_mel

HIFI_FLOP_PER_FRAME = 614.11e6          # SURVEY.md §8d (verified with torch.utils.flop_counter on the reference)
# ... and its split (MFLOP per mel frame): conv_pre, the four upsamplers, the four MRF stages, conv_post
STAGE_MFLOP = {"conv_pre": 0.573, "ups0": 4.194, "ups1": 8.389, "ups2": 4.194, "ups3": 2.097,
               "mrf0": 132.12, "mrf1": 264.24, "mrf2": 132.12, "mrf3": 66.06, "conv_post": 0.115}
# rocprofv3 symbols of each stage's kernels (profiles/r*_pmc_traffic_hifi.json is keyed by symbol): (substring, launches per forward)
STAGE_KERNELS = {"mrf0": [("conv_pair256_kernel", 9)],
                 # round 6: the weights-stationary pair kernel <C, K, dilation, MRF mode> (csrc/pairws.hip): the C = 64 stage's nine launches and the k = 3
                 # block of the C = 128 stage (k = 7 / 11 there stay on conv_pair_kernel)
                 "mrf1": [("conv_pair_kernel", 6), ("pair_ws_kernel<128, 3, 1, 0", 1), ("pair_ws_kernel<128, 3, 3, 0", 1), ("pair_ws_kernel<128, 3, 5, 0", 1)],
                 "mrf2": [("pair_ws_kernel<64, %d, %d, %d" % (k, d, m), 1) for k, m5 in ((3, 0), (7, 1), (11, 2)) for d, m in ((1, 0), (3, 0), (5, m5))],
                 "mrf3": [("resblock1_kernel<32, 3", 1), ("resblock1_kernel<32, 7", 1), ("resblock1_kernel<32, 11", 1), ("mrf32_post_kernel", 1)],
                 "ups0": [("win_conv_kernel<512, 128", 1)], "ups1": [("win_conv_kernel<256, 224", 1)],
                 "ups2": [("ups2_kernel<128, 64", 1)], "ups3": [("ups2_kernel<64, 32", 1)], "conv_post": [("conv_post_kernel", 1)]}
# tensor elements per mel frame at each stage's resolution (channels x frames-per-mel-frame), 2 bytes each: the algorithmic
# minimum of a fused MRF stage is one read of its input and one write of its output
STAGE_ELEMS = {"mrf0": 256 * 8, "mrf1": 128 * 64, "mrf2": 64 * 128, "mrf3": 32 * 256}


def _pmc_hifi():
    """kernel symbol -> HBM bytes per launch from the newest committed profiles/r*_pmc_traffic_hifi.json (tools/pmc_hifi.sh:
    separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes; FETCH_SIZE doubled per the gfx950 correction)."""
    import glob
    import json
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for f in sorted(glob.glob(os.path.join(root, "profiles", "r*_pmc_traffic_hifi.json")), reverse=True):
        try:
            doc = json.load(open(f))
        except (OSError, ValueError):
            continue
        from .lib import source_fingerprint
        return ({r["kernel"]: r["hbm_bytes_per_launch"] for r in doc.get("kernels", [])}, os.path.relpath(f, root),
                doc.get("csrc_fingerprint") != source_fingerprint())
    return {}, None, False


def stage_rooflines(gen, mel, B, T, iters=5):
    """"report the dilated-conv kernels against both bounds".  Per generator stage: device time between HIP events
    recorded at the stage boundaries on the launch stream (eager launches), algorithmic FLOPs -> fraction of the f16 MFMA peak,
    and HBM bytes — the PMC counters of the stage's kernels where a committed summary has them, and the algorithmic minimum
    (one read + one write of the stage tensor) — -> fraction of the 8 TB/s HBM peak.  A stage well below BOTH is bound by
    neither (per-workgroup latency phases)."""
    frames = B * T
    acc = {}
    for _ in range(iters):
        gen._stage_marks = marks = []
        try:
            gen(mel)
        finally:
            gen._stage_marks = None
        torch.cuda.synchronize()
        for (n0, e0), (n1, e1) in zip(marks[:-1], marks[1:]):
            acc.setdefault(n1, []).append(e0.elapsed_time(e1))
    pmc, src, stale = _pmc_hifi()
    out = {}
    for name, ms_list in acc.items():
        ms = sorted(ms_list)[len(ms_list) // 2]
        flops = STAGE_MFLOP.get(name, 0.0) * 1e6 * frames
        rec = {"ms": ms, "tflops": flops / (ms * 1e-3) / 1e12, "mfma_roofline_frac": flops / (ms * 1e-3) / 2.5e15}
        if name in STAGE_ELEMS:
            alg = 2.0 * STAGE_ELEMS[name] * 2 * frames
            rec["hbm_bytes_algorithmic"] = alg
            rec["hbm_roofline_frac_algorithmic"] = alg / (ms * 1e-3) / 8e12
        if name in STAGE_KERNELS and pmc:
            tot, found = 0.0, True
            for sub, n in STAGE_KERNELS[name]:
                hit = [v for k, v in pmc.items() if sub in k]
                if hit:
                    tot += hit[0] * n
            if tot > 0:
                rec["hbm_bytes_pmc"] = tot
                rec["hbm_roofline_frac"] = tot / (ms * 1e-3) / 8e12
                rec["pmc_source"] = src
                rec["profile_stale"] = stale      # the kernel sources changed after the PMC passes were taken (lib.source_fingerprint)
        out[name] = rec
    return out


def build_generator(cfg, dev, seed=1234):
    g = Generator(cfg.hifi)
    g.reset_parameters(seed)
    g.to(dev)
    g.remove_weight_norm()
    return g.eval()


def hifi_rtf(cfg, dev, B=8, T=384, iters=20, warmup=3, use_graph=True, warm_replays=20):
    """`warm_replays`: untimed replays of the captured graph before the timed ones — like the W warmup steps of bench.py's train-step leg, which
    are replays too.  (Round 5: the leg timed replays 2-21 of a freshly captured graph and read 2.44 ms where replays 6-55 of the same graph read
    2.34: the graph's buffers are new memory at capture, and the first replays pay for that — tools/debug/hifi_warm_curve.py: 2.587 ms over the
    first ten, 2.39, then 2.38 steadily.)"""
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
        for _ in range(max(1, warm_replays)):
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
    ops.to_host(ops.to_int16(wav, float(cfg.hifi.MAX_WAV_VALUE)))            # (first call: allocates the pinned staging buffer)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    i16 = ops.to_host(ops.to_int16(wav, float(cfg.hifi.MAX_WAV_VALUE)))
    d2h = time.perf_counter() - t1
    flops = HIFI_FLOP_PER_FRAME * B * T
    return {"workload": "HiFi-GAN V1 generator, B=%d, T=%d mel frames -> %d samples @ %d Hz (BASELINE.json configs[2])" % (B, T, samples, int(sr)),
            "rtf": wall / audio_s, "ms_per_batch": 1e3 * wall, "device_ms_per_batch": 1e3 * dev_s, "samples_per_s": samples / wall,
            "audio_seconds_per_batch": audio_s, "tflops": flops / wall / 1e12, "mfma_roofline_frac": flops / wall / 2.5e15,
            "int16_d2h_ms": 1e3 * d2h, "launch": "hipGraph replay" if use_graph else "eager", "dtype": "f16 (fp32 accumulate)",
            "iters": iters, "warm_replays": max(1, warm_replays) if use_graph else 0, "stages": stage_rooflines(gen, mel, B, T)}
