#!/usr/bin/env python3
"""bench.py — FS2 train-step throughput (BASELINE.json configs[1]) on N MI355X GPUs of one node.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

A step = one full `main_train_step` (forward, loss, backward, [RCCL gradient all-reduce], global-norm clip, Adam with the
reference LR schedule, zero_grad; grad_acc_step = 1) on one synthetic batch per GPU: B=16 utterances x L=64 phonemes,
durations 1..11 frames -> T_max ~423 mel frames, 80-bin mel, 65 speakers, dropout ON, bf16 MFMA compute with fp32
accumulation and fp32 master weights.  Inputs are resident in HBM before the timed region.  `value` counts VALID
mel frames (sum of mel_lens) over all ranks per second.  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
from tts_king_amd import switches  # noqa: E402

PEAK_MFMA_BF16_TFLOPS = 2500.0      # dense bf16 MFMA peak, /opt/skills/guides/MI355X_MICROARCH.md
PEAK_HBM_GBS = 8000.0


def fs2_flops_per_step(B, L, T):
    """Exact train-step FLOPs (3 x forward; SURVEY.md §8d, verified with torch.utils.flop_counter on the reference)."""
    fwd = B * (25429504 * L + 4096 * L * L + 43327488 * T + 6144 * T * T)
    return 3.0 * fwd


def host_threads():
    """Threads for the CPU baseline: the cores this process may actually use (a 1-GPU box gives a 16-core share of a much
    larger host through its cgroup quota; os.cpu_count() and the affinity mask say 256), at most TTSK_CPU_THREADS."""
    from tts_king_amd.hostcpu import host_cores
    return host_cores(cap=int(switches.get("TTSK_CPU_THREADS")))


def cpu_baseline(cfg, B, L, n_steps=5):
    """The oracle (CPU fp32 restatement of the reference step) timed on this box's host cores, same workload."""
    import copy
    from oracle import fs2 as ofs2
    from tts_king_amd.fastspeech2 import FastSpeech2
    from tts_king_amd.synthetic import make_batch
    c = copy.deepcopy(cfg)
    c.train_config["optimizer"]["grad_acc_step"] = 1
    torch.set_num_threads(host_threads())
    m = FastSpeech2(c.preprocess_config, c.model_config, 65, device="cpu", seed=1234)
    sd = {k: v.detach().clone().contiguous() for k, v in m.state_dict().items()}
    tr = ofs2.OracleTrainer(sd, c.model_config, c.train_config, 0)
    b = make_batch(B, L, seed=1234)
    t0 = time.perf_counter()
    tr.train_step(b, 1)                                  # warm-up
    warm = time.perf_counter() - t0
    n_steps = max(1, min(n_steps, int(20.0 / max(warm, 1e-3))))      # bounded sample: ~20 s of CPU work
    t0 = time.perf_counter()
    for s in range(n_steps):
        tr.train_step(b, s + 2)
    dt = (time.perf_counter() - t0) / n_steps
    frames = int(b[7].sum())
    return {"value": frames / dt, "unit": "mel-frames/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": "%d full train steps (dropout on) of the same B=%d, L=%d batch, fp32, %.2f s/step" % (n_steps, B, L, dt)}


def hifi_cpu_baseline(cfg, B=8, T=384, runs=3):
    """The oracle (CPU fp32 restatement of the reference generator) timed on this box's host cores."""
    import os
    from oracle import hifigan as ohifi
    torch.set_num_threads(host_threads())
    from tts_king_amd.hifigan import Generator
    from tts_king_amd.synthetic import make_mel
    g = Generator(cfg.hifi)
    g.reset_parameters(1234)
    sd = ohifi.fold_weight_norm({k: v.detach().clone() for k, v in g.state_dict().items()})
    mel = make_mel(B, T, seed=1234)
    with torch.no_grad():
        ohifi.generator(sd, cfg.hifi, mel[:1, :, :32])          # warm-up
        times = []
        for _ in range(runs):
            t0 = time.perf_counter()
            ohifi.generator(sd, cfg.hifi, mel)
            times.append(time.perf_counter() - t0)
        dt = sorted(times)[len(times) // 2]                       # median
    audio_s = B * T * 256 / float(cfg.hifi.sampling_rate)
    return {"rtf": dt / audio_s, "seconds_per_batch": dt, "cores": torch.get_num_threads(), "kind": "port",
            "sample": "median of %d runs of the same B=%d, T=%d batch, fp32" % (runs, B, T)}


def mel_extraction_leg(cfg, dev, B=16, T=423, iters=20, with_cpu=True):
    """SURVEY §8 row f-3 beside the two headline workloads: log-mel + energy of the FS2 batch's audio (B utterances of
    T frames, 22.05 kHz) through tts_king_amd/audio.py, device time by HIP events on the launch stream, and the oracle
    (torch.stft, CPU fp32) on a bounded sample.  FLOPs: the three hi/lo STFT contractions, 3 * 2 * frames * 1032 * 1024."""
    from tts_king_amd.audio import TacotronSTFT
    p = cfg.preprocess_config["preprocessing"]
    n_fft, hop, win = p["stft"]["filter_length"], p["stft"]["hop_length"], p["stft"]["win_length"]
    n_mel, sr, fmin, fmax = p["mel"]["n_mel_channels"], p["audio"]["sampling_rate"], p["mel"]["mel_fmin"], p["mel"]["mel_fmax"]
    g = torch.Generator().manual_seed(1234)
    y = (torch.rand(B, (T - 1) * hop, generator=g) * 2 - 1) * 0.5
    stft = TacotronSTFT(n_fft, hop, win, n_mel, sr, fmin, fmax, device=dev)
    yd = y.to(dev)
    for _ in range(3):
        stft._ex(yd)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        mel, energy = stft._ex(yd)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    frames = B * mel.shape[2]
    flops = 3 * 2.0 * B * (mel.shape[2] + 3) * 1032 * n_fft
    rec = {"workload": "log-mel + energy, B=%d x %d frames (n_fft %d, hop %d, %d mels, %d Hz)" % (B, mel.shape[2], n_fft, hop, n_mel, sr),
           "frames_per_s": frames / (ms * 1e-3), "ms_per_batch": ms, "audio_seconds_per_batch": B * y.shape[1] / float(sr),
           "tflops": flops / (ms * 1e-3) / 1e12, "dtype": "fp16 hi/lo split operands, fp32 accumulate", "launch": "eager, 3 kernels"}
    if with_cpu:
        from oracle import audio as oaudio
        torch.set_num_threads(host_threads())
        oaudio.tacotron_mel(y[:1], n_fft, hop, win, n_mel, sr, fmin, fmax)
        t0 = time.perf_counter()
        ref_mel, _ = oaudio.tacotron_mel(y, n_fft, hop, win, n_mel, sr, fmin, fmax)
        dt = time.perf_counter() - t0
        rec["max_abs_vs_oracle"] = float((mel.cpu() - ref_mel).abs().max())
        rec["cpu_baseline"] = {"value": frames / dt, "unit": "frames/s", "cores": torch.get_num_threads(), "kind": "port",
                               "sample": "1 run of the same batch (conv-STFT oracle, fp32), %.3f s" % dt}
    return rec


def e2e_synth_leg(cfg, dev, L=64, iters=20, with_cpu=True):
    """BASELINE.json configs[4]: one utterance phoneme ids -> mel -> int16 waveform on the host, through the replayed
    hipGraphs of tts_king_amd/synth.py (what `TTSKing.generate_mel` + `mel_to_wav` run with `mi355x.hip_graph: true`) and
    through plain launches.  Synthetic weights predict zero durations, so the duration head's bias is set to log(7.6):
    64 phonemes -> ~420 frames, the training batch's utterance length."""
    import math
    import numpy as np
    from tts_king_amd import ops
    from tts_king_amd.fastspeech2 import FastSpeech2
    from tts_king_amd.hifi_bench import build_generator
    from tts_king_amd.synth import GraphedSynthesizer
    model = FastSpeech2(cfg.preprocess_config, cfg.model_config, 65, device=dev, seed=1234).eval()
    sd = model.state_dict()
    sd["variance_adaptor.duration_predictor.linear_layer.bias"].fill_(math.log(7.6))
    sd["variance_adaptor.duration_predictor.linear_layer.weight"].zero_()
    model.load_state_dict(sd)
    gen = build_generator(cfg, dev)
    synth = GraphedSynthesizer(model, gen)
    g = torch.Generator().manual_seed(1234)
    texts = torch.randint(1, 200, (1, L), generator=g).to(dev)
    spk = torch.zeros(1, dtype=torch.int64, device=dev)
    scale = float(cfg.hifi.MAX_WAV_VALUE)

    def graphed():
        post, lens = synth.mel(spk, texts)
        wav = synth.wav(post.transpose(1, 2))
        return ops.to_host(ops.to_int16(wav, scale)), int(lens[0])      # (as hifiapi.HIFIapi.generate hands the samples over)

    def eager():
        with torch.no_grad():
            sl = torch.full((1,), L, dtype=torch.int64, device=dev)
            x3, dur, total, _ = model.eval_front(spk, texts, sl, L, 1.0, 1.0, 1.0)
            T = max(int(total.max().item()), 1)
            _, post, lens, _ = model.eval_back(x3, dur, L, T)
            wav = gen(post.transpose(1, 2).contiguous())
            return ops.to_host(ops.to_int16(wav, scale)), int(lens[0])

    out = {}
    for name, fn in (("hipgraph", graphed), ("eager", eager)):
        for _ in range(3):
            pcm, T = fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(iters):
            pcm, T = fn()
        out[name] = (time.perf_counter() - t0) / iters
    audio_s = T * 256 / float(cfg.hifi.sampling_rate)
    rec = {"workload": "tts_king synth, 1 utterance: %d phonemes -> %d mel frames -> %d int16 samples on the host (BASELINE.json configs[4])" % (L, T, pcm.shape[-1]),
           "latency_ms": 1e3 * out["hipgraph"], "latency_ms_eager": 1e3 * out["eager"], "rtf": out["hipgraph"] / audio_s,
           "audio_seconds": audio_s, "launch": "3 replayed hipGraphs + 1 host read of the frame count + int16 D2H"}
    if with_cpu:
        from oracle import fs2 as ofs2, hifigan as ohifi
        torch.set_num_threads(host_threads())
        sd_cpu = {k: v.detach().float().cpu().clone() for k, v in model.state_dict().items()}
        hsd = ohifi.fold_weight_norm({k: v.detach().float().cpu().clone() for k, v in gen.state_dict().items()}) if any(
            k.endswith("weight_g") for k in gen.state_dict()) else {k: v.detach().float().cpu().clone() for k, v in gen.state_dict().items()}
        with torch.no_grad():
            t0 = time.perf_counter()
            res = ofs2.fs2_forward(sd_cpu, cfg.model_config, spk.cpu(), texts.cpu(), torch.tensor([L]), L, train=False)
            wav = ohifi.generator(hsd, cfg.hifi, res[9].transpose(1, 2))
            dt = time.perf_counter() - t0
        rec["cpu_baseline"] = {"value": 1e3 * dt, "unit": "ms per utterance", "cores": torch.get_num_threads(), "kind": "port",
                               "sample": "1 utterance, fp32 oracle FS2 eval + HiFi-GAN", "rtf": dt / audio_s}
    return rec


def _profile_docs(pattern):
    """(doc, relative path) of the committed PMC summaries matching profiles/<pattern>, newest round first."""
    import glob
    here = os.path.dirname(os.path.abspath(__file__))
    for f in sorted(glob.glob(os.path.join(here, "profiles", pattern)), reverse=True):
        try:
            yield json.load(open(f)), os.path.relpath(f, here)
        except (OSError, ValueError):
            continue


def profile_is_stale(doc):
    """A committed profile is stale when the kernel sources changed after it was taken: its `csrc_fingerprint` (tools/pmc_summary.py,
    tts_king_amd/lib.py:source_fingerprint) differs from this tree's.  Profiles of earlier rounds carry no fingerprint: stale."""
    from tts_king_amd.lib import source_fingerprint
    return doc.get("csrc_fingerprint") != source_fingerprint()


def pmc_traffic(symbol):
    """HBM bytes per launch of the kernels whose symbol contains `symbol` (dispatch-weighted mean) from the newest committed PMC
    summary (profiles/r*_pmc_traffic.json, written by tools/pmc_bench.sh + tools/pmc_summary.py from separate rocprofv3 --pmc
    FETCH_SIZE / WRITE_SIZE passes over this same bench command; counters cannot be collected from inside the timed process).
    None when there is no summary for it."""
    for doc, rel in _profile_docs("r*_pmc_traffic.json"):
        rows = [r for r in doc.get("kernels", []) if symbol in r["kernel"]]
        if rows:
            n = sum(r["dispatches"] for r in rows)
            mean = lambda key: sum(r[key] * r["dispatches"] for r in rows) / n
            return {"hbm_bytes_per_launch": mean("hbm_bytes_per_launch"), "stale": profile_is_stale(doc),
                    "source": "%s: FETCH_SIZE x2 %.1f MB + WRITE_SIZE %.1f MB per launch, %d dispatches of %d symbol(s)" % (
                        rel, mean("fetch_bytes_per_launch") / 1e6, mean("write_bytes_per_launch") / 1e6, n, len(rows))}
    return None


def pmc_mfma_busy(symbol, section="fs2_train_step"):
    """mfma_busy_frac (cycle-weighted over the symbols containing `symbol`) from the newest committed profiles/r*_mfma_util.json
    (tools/pmc_mfma.sh: a separate rocprofv3 --pmc pass over this bench command).  None when there is no summary."""
    for doc, rel in _profile_docs("r*_mfma_util.json"):
        rows = [r for r in doc.get(section, []) if symbol in r["kernel"] and r.get("mfma_busy_frac") is not None]
        if rows:
            w = [r["avg_cycles"] * r["dispatches"] for r in rows]
            return {"mfma_busy_frac": sum(r["mfma_busy_frac"] * wi for r, wi in zip(rows, w)) / sum(w),
                    "lds_conflict_frac": sum((r.get("lds_conflict_frac") or 0.0) * wi for r, wi in zip(rows, w)) / sum(w),
                    "stale": profile_is_stale(doc), "source": rel}
    return None


def profile_family_time(symbol):
    """Device time per step of the kernels whose symbol contains `symbol`, from the newest committed
    profiles/r*_bench_kernel_stats.csv (rocprofv3 --kernel-trace --stats of this same bench command, tools/round_profiles.sh).
    Steps traced = the calls of the once-per-step Adam launch (or the sidecar's `steps`).  The CSV carries no fingerprint of its own:
    the sidecar profiles/<tag>_bench_kernel_stats.meta.json (tools/install_profiles.py) or, for summaries installed before it existed,
    the PMC summary of the same round_profiles.sh run (profiles/<tag>_pmc_traffic.json) says which sources it was measured on.
    None when there is no summary with such kernels."""
    import csv
    import glob
    here = os.path.dirname(os.path.abspath(__file__))
    for f in sorted(glob.glob(os.path.join(here, "profiles", "r*_bench_kernel_stats.csv")), reverse=True):
        try:
            rows = list(csv.DictReader(open(f)))
        except (OSError, ValueError):
            continue
        tag = os.path.basename(f)[:-len("_bench_kernel_stats.csv")]
        meta = {}
        for side in ("%s_bench_kernel_stats.meta.json" % tag, "%s_pmc_traffic.json" % tag):
            try:
                meta = json.load(open(os.path.join(here, "profiles", side)))
                break
            except (OSError, ValueError):
                continue
        steps = meta.get("steps") or sum(int(r["Calls"]) for r in rows if "adam_pack_kernel" in r["Name"] or "adam_clip_kernel" in r["Name"])
        hit = [r for r in rows if symbol in r["Name"]]
        if not hit or not steps:
            continue
        calls = sum(int(r["Calls"]) for r in hit)
        ns = sum(float(r["TotalDurationNs"]) for r in hit)
        from tts_king_amd.lib import source_fingerprint
        return {"ms_per_step": ns / steps * 1e-6, "launches_per_step": calls / steps, "avg_launch_us": ns / calls * 1e-3, "steps_traced": steps,
                "symbols": len(hit), "source": os.path.relpath(f, here), "stale": meta.get("csrc_fingerprint") != source_fingerprint()}
    return None


# kernel families of the train step: trace kind (tts_king_amd/ops.py: GEMM_TRACE entries) -> (family, rocprofv3 symbol substring, source)
FAMILIES = {"win_conv": ("win_conv_kernel", "tts_king_amd/csrc/ffn_conv.hip"), "win_ln": ("win_ln_kernel", "tts_king_amd/csrc/gemm_ln.hip"),
            "ln_bwd_proj": ("ln_bwd256_proj_kernel", "tts_king_amd/csrc/layernorm.hip"), "flash_attention": ("flash_", "tts_king_amd/csrc/flash_attn.hip"),
            "dwconv": ("dwconv_kernel", "tts_king_amd/csrc/dwconv.hip"), "dwgemm": ("dwgemm_kernel", "tts_king_amd/csrc/dwgemm.hip"),
            "gemm": ("gemm", "tts_king_amd/csrc/gemm.hip, gemm2.hip"), "batchnorm": ("bn_", "tts_king_amd/csrc/batchnorm.hip"),
            "clip_adam": ("adam_", "tts_king_amd/csrc/optim.hip")}


def _family_of(kind):
    if kind in FAMILIES:
        return kind
    if kind.startswith("dwconv"):
        return "dwconv"
    return "gemm"            # NT / TT / grouped tile configurations of ttsk_gemm


def step_roofline(enqueue, batch, steps=3):
    """Roofline of the train step by kernel FAMILY.  Every launch of the MFMA families (window convs, fused projection + LayerNorm,
    LayerNorm-backward + projection, flash attention, dwconv, dwgemm, the ttsk_gemm tiles) and of the two HBM-bound passes with real
    time share (BatchNorm, clip + Adam) in `steps` eager train steps is bracketed by HIP events on its launch stream
    (tts_king_amd/ops.py: GEMM_TRACE / _family); per family: launches, device time, algorithmic FLOPs (2*M*N*K*taps: the reference's
    counts) -> fraction of the dense bf16 MFMA peak, and the HBM bytes per step by PMC where a committed summary has the
    family's kernels.  The headline `achieved` / `frac` are the family that takes the most TIME; `dominant_by_flops` is reported
    beside it.  Eager brackets: a launch's bracket can include the tail of a kernel on another stream, never another launch of
    its own stream."""
    from tts_king_amd import ops
    enqueue(batch)
    torch.cuda.synchronize()
    trace = []
    ops.GEMM_TRACE = trace
    try:
        for _ in range(steps):
            # Park the GPU behind a spin kernel while the host enqueues the whole step (~4 ms of Python): the launches then run back to
            # back, as in the replayed graph, and a bracket holds its kernel — not the host time between recording its first event
            # and enqueueing the launch (eager launches are host-bound: brackets measured 10-50 % long without this).
            torch.cuda._sleep(int(60e6))
            enqueue(batch)
            torch.cuda.synchronize()
    finally:
        ops.GEMM_TRACE = None
    fam = {}
    for e0, e1, fl, kind, shape in trace:
        d = fam.setdefault(_family_of(kind), [0.0, 0.0, 0])
        d[0] += e0.elapsed_time(e1); d[1] += fl; d[2] += 1
    rows = []
    stale = False
    for name, (ms, fl, n) in fam.items():
        sym, src = FAMILIES[name]
        tr, mb = pmc_traffic(sym), pmc_mfma_busy(sym)
        stale = stale or bool(tr and tr["stale"]) or bool(mb and mb["stale"])
        tf = fl / (ms * 1e-3) / 1e12 if ms > 0 else 0.0
        rows.append({"family": name, "kernels": "%s* (%s)" % (sym, src), "launches": n // steps, "ms_per_step": ms / steps,
                     "gflop": fl / steps / 1e9, "tflops": tf, "frac_of_peak": tf / PEAK_MFMA_BF16_TFLOPS,
                     "hbm_bytes_pmc": tr["hbm_bytes_per_launch"] * (n // steps) if tr else None,
                     "hbm_frac_of_peak": (tr["hbm_bytes_per_launch"] * n / (ms * 1e-3) / 8e12) if (tr and ms > 0) else None,
                     "mfma_busy_frac": mb["mfma_busy_frac"] if mb else None, "lds_conflict_frac": mb["lds_conflict_frac"] if mb else None,
                     "pmc_source": (tr or mb or {}).get("source")})
    rows.sort(key=lambda r: -r["ms_per_step"])
    top = rows[0]
    # the dominant family launch by launch: its instances by (wrapper, algorithmic GFLOP) — the family figure averages w_1's 32 GFLOP
    # launches with the 80-channel and phoneme-side ones
    inst = {}
    for e0, e1, fl, kind, shape in trace:
        if _family_of(kind) == top["family"]:
            d = inst.setdefault((shape[6] if len(shape) > 6 else kind, round(fl / 1e9, 2)), [0.0, 0])
            d[0] += e0.elapsed_time(e1); d[1] += 1
    inst_rows = [{"op": k[0], "gflop_per_launch": k[1], "launches": n // steps, "us_per_launch": 1e3 * ms / n,
                  "tflops": k[1] / (ms / n) if ms > 0 else 0.0,
                  "frac_of_peak": (k[1] / (ms / n)) / PEAK_MFMA_BF16_TFLOPS if ms > 0 else 0.0} for k, (ms, n) in inst.items()]
    inst_rows.sort(key=lambda r: -r["us_per_launch"] * r["launches"])
    byfl = max(rows, key=lambda r: r["gflop"])
    tot_ms = sum(r["ms_per_step"] for r in rows)
    tot_gf = sum(r["gflop"] for r in rows)
    mfma = top["gflop"] > 0
    tr = pmc_traffic(FAMILIES[top["family"]][0])
    rec = {"bound": "mfma" if mfma else "hbm",
           "achieved": top["tflops"] if mfma else (top["hbm_bytes_pmc"] or 0.0) / (top["ms_per_step"] * 1e-3) / 1e9,
           "peak": PEAK_MFMA_BF16_TFLOPS if mfma else 8000.0, "unit": "TFLOP/s" if mfma else "GB/s",
           "kernel": top["kernels"], "family": top["family"], "launches_per_step": top["launches"],
           "avg_launch_us": 1e3 * top["ms_per_step"] / max(top["launches"], 1), "avg_launch_gflop": top["gflop"] / max(top["launches"], 1),
           "kernel_ms_per_step": top["ms_per_step"],
           "traffic": tr["hbm_bytes_per_launch"] if tr else None, "traffic_source": tr["source"] if tr else None,
           "mfma_busy_frac": top["mfma_busy_frac"],
           "dominant_by_time": {k: top[k] for k in ("family", "launches", "ms_per_step", "gflop", "tflops", "frac_of_peak")},
           "dominant_by_flops": {k: byfl[k] for k in ("family", "launches", "ms_per_step", "gflop", "tflops", "frac_of_peak", "mfma_busy_frac")},
           "families": rows[:6], "instances": inst_rows[:8],
           "traced": {"launches_per_step": len(trace) // steps, "ms_per_step": tot_ms, "gflop_per_step": tot_gf,
                      "tflops": tot_gf / tot_ms if tot_ms > 0 else 0.0},
           "profile_stale": stale}
    # `achieved` / `frac`: the family's algorithmic FLOPs per step (counted live, above) over its device time per step in the committed
    # rocprofv3 --stats summary of this bench command, when that summary was measured on these kernel sources; the live HIP-event
    # brackets (eager launches behind a spin kernel: 3-4 us of event and launch-boundary time per bracket) are reported beside it and
    # stand in when there is no current summary.
    rec["achieved_event_brackets"] = rec["achieved"]
    rec["frac_event_brackets"] = rec["achieved"] / rec["peak"]
    rec["frac_source"] = "live HIP-event brackets"
    prof = profile_family_time(FAMILIES[top["family"]][0]) if mfma else None
    if prof is not None:
        rec["profile"] = prof
        rec["profile_stale"] = rec["profile_stale"] or prof["stale"]
        if not prof["stale"]:
            rec["achieved"] = top["gflop"] / prof["ms_per_step"]
            rec["avg_launch_us"] = prof["avg_launch_us"]
            rec["kernel_ms_per_step"] = prof["ms_per_step"]
            rec["frac_source"] = "%s: %d symbols, %.1f launches and %.3f ms per step over %d traced steps" % (
                prof["source"], prof["symbols"], prof["launches_per_step"], prof["ms_per_step"], prof["steps_traced"])
    rec["frac"] = rec["achieved"] / rec["peak"]
    return rec


LINE_LIMIT = 6144          # the driver keeps the last 8 KB of stdout: the whole line must fit with room to spare


def _sig(x, n=6):
    """Floats at n significant digits (the line is read by people and a tail-limited parser; the full precision is in the side file)."""
    if isinstance(x, float):
        return float("%.*g" % (n, x))
    if isinstance(x, dict):
        return {k: _sig(v, n) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_sig(v, n) for v in x]
    return x


def compact_record(rec, full_path=None):
    """The ONE line bench.py prints: the contract's keys, `config`, `roofline` (dominant family only), `cpu_baseline` and a compact
    `hifi_gan` (the metric's second half), plus one scalar per auxiliary leg.  Everything else (`roofline.families` / `instances`, the
    HiFi-GAN stages' byte counts, the auxiliary legs' records) goes to the side file and to stderr.  Sections are dropped from the
    end of `optional` until the line fits LINE_LIMIT; the headline halves are never dropped."""
    keep = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
            "data", "config", "mel_frames_per_s_per_gpu", "model_tflops", "step_mfma_roofline_frac", "final_losses", "unknown_switches")
    out = {k: rec[k] for k in keep if k in rec}
    h = rec.get("hifi_gan")
    if h:
        hk = ("rtf", "ms_per_batch", "device_ms_per_batch", "tflops", "mfma_roofline_frac", "warm_replays", "iters", "int16_d2h_ms",
              "samples_per_s", "launch", "dtype", "workload", "cpu_baseline")
        out["hifi_gan"] = {k: h[k] for k in hk if k in h}
        if h.get("stages"):
            out["hifi_gan"]["stages"] = {n: {"ms": s["ms"], "mfma_roofline_frac": s["mfma_roofline_frac"],
                                             **({"hbm_bytes_pmc": s["hbm_bytes_pmc"]} if "hbm_bytes_pmc" in s else {})}
                                         for n, s in h["stages"].items() if n.startswith("mrf")}
    r = rec.get("roofline")
    if r:
        rk = ("bound", "achieved", "peak", "unit", "frac", "frac_source", "achieved_event_brackets", "frac_event_brackets", "traffic",
              "traffic_source", "kernel", "family", "launches_per_step", "avg_launch_us", "avg_launch_gflop", "kernel_ms_per_step",
              "mfma_busy_frac", "profile_stale")
        out["roofline"] = {k: r[k] for k in rk if k in r}
        out["roofline"]["families_ms_per_step"] = {f["family"]: f["ms_per_step"] for f in r.get("families", [])}
    if "cpu_baseline" in rec:
        out["cpu_baseline"] = rec["cpu_baseline"]
    aux = {k: rec[k] for k in ("ms_per_step_eager", "ms_per_step_grad_acc4", "ms_per_step_train_loop") if k in rec}
    dp = rec.get("dp_schedule_1gpu") or {}
    if dp.get("by_schedule"):
        aux["dp1_ms_per_step"] = {k: v.get("dp1_reducer_ms_per_step") for k, v in dp["by_schedule"].items()}
    if rec.get("e2e_synth"):
        aux["e2e_synth_latency_ms"] = rec["e2e_synth"].get("latency_ms")
        aux["e2e_synth_cpu_ms"] = (rec["e2e_synth"].get("cpu_baseline") or {}).get("value")
    if rec.get("mel_extraction"):
        aux["mel_extraction_ms_per_batch"] = rec["mel_extraction"].get("ms_per_batch")      # (NOT the HiFi-GAN figure: that is hifi_gan.ms_per_batch)
    if aux:
        out["aux"] = aux
    if full_path:
        out["full_record"] = full_path
    out = _sig(out)
    optional = [("aux",), ("hifi_gan", "stages"), ("roofline", "families_ms_per_step"), ("final_losses",), ("roofline", "traffic_source"),
                ("hifi_gan", "workload")]
    line = json.dumps(out)
    while len(line) > LINE_LIMIT and optional:
        path = optional.pop(0)
        d = out
        for k in path[:-1]:
            d = d.get(k, {})
        d.pop(path[-1], None)
        line = json.dumps(out)
    if len(line) > LINE_LIMIT:
        raise RuntimeError("bench line is %d bytes (limit %d) after dropping every optional section" % (len(line), LINE_LIMIT))
    return line


def write_full_record(rec):
    """The long form next to the line: gpurun_out/bench_full.json under the repo (merged back by gpurun), else the temp directory.
    Returns the path written (relative to the repo when inside it), or None."""
    import tempfile
    for d in (os.path.join(ROOT, "gpurun_out"), tempfile.gettempdir()):
        try:
            os.makedirs(d, exist_ok=True)
            f = os.path.join(d, "bench_full.json")
            with open(f, "w") as fh:
                json.dump(rec, fh, indent=1)
            return os.path.relpath(f, ROOT) if f.startswith(ROOT) else f
        except OSError:
            continue
    return None


def _time_loop(fn, n, sync=True):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / n


def extra_train_legs(cfg, dev, B, L, steps=40):
    """The numbers beside the headline that VERDICT r01 asked for (1 GPU): the same step launched eagerly; the reference's default
    `grad_acc_step: 4` cycle (three accumulate-only micro-steps + one update, two replayed graphs); and the TRAINER's loop —
    tts_king_amd.engine.TrainEngine fed by DeviceFeeder with varying-T synthetic batches (shape buckets, graph cache, pinned H2D),
    i.e. what `python train.py` runs per step."""
    import copy
    from tts_king_amd.dataset import DeviceFeeder
    from tts_king_amd.engine import TrainEngine
    from tts_king_amd.fastspeech2 import FastSpeech2
    from tts_king_amd.graph import GraphedTrainStep, make_enqueue
    from tts_king_amd.loss import FastSpeech2Loss
    from tts_king_amd.optimizer import ScheduledOptim
    from tts_king_amd.synthetic import make_batch
    from tts_king_amd.train_step import to_device
    out = {}
    loss_fn = FastSpeech2Loss(cfg.preprocess_config, cfg.model_config)
    # ---- eager launches of the headline step
    c1 = copy.deepcopy(cfg)
    c1.train_config["optimizer"]["grad_acc_step"] = 1
    model = FastSpeech2(c1.preprocess_config, c1.model_config, 65, device=dev, seed=1234).train()
    opt = ScheduledOptim(model, c1.train_config, c1.model_config, 0)
    batch = to_device(make_batch(B, L, seed=1234), dev)
    enq = make_enqueue(model, opt, c1, loss_fn)
    for _ in range(3):
        enq(batch)
    out["ms_per_step_eager"] = _time_loop(lambda: enq(batch), 20)
    # ---- grad_acc_step = 4: micro-steps 1-3 accumulate, the 4th also clips / updates (reference default, config.yaml:51)
    c4 = copy.deepcopy(cfg)
    c4.train_config["optimizer"]["grad_acc_step"] = 4
    m4 = FastSpeech2(c4.preprocess_config, c4.model_config, 65, device=dev, seed=1234).train()
    o4 = ScheduledOptim(m4, c4.train_config, c4.model_config, 0)
    # the first micro-step after an update OVERWRITES the gradient buffer (the update does not zero it), the later ones accumulate
    g_first = GraphedTrainStep(make_enqueue(m4, o4, c4, loss_fn, step_is_update=False, accumulate=False), batch, warmup=2)
    g_acc = GraphedTrainStep(make_enqueue(m4, o4, c4, loss_fn, step_is_update=False, accumulate=True), batch, warmup=0)
    g_upd = GraphedTrainStep(make_enqueue(m4, o4, c4, loss_fn, step_is_update=True, accumulate=True), batch, warmup=0)

    def cycle():
        g_first.run(); g_acc.run(); g_acc.run(); g_upd.run()
    for _ in range(3):
        cycle()
    ms_cycle = _time_loop(cycle, max(5, steps // 4))
    out["ms_per_step_grad_acc4"] = ms_cycle / 4.0
    out["grad_acc4"] = {"ms_per_optimizer_update": ms_cycle, "micro_steps_per_update": 4, "graphs": "overwrite + 2 x accumulate + accumulate-and-update",
                        "valid_mel_frames_per_s": 4 * int(batch[7].sum()) / (ms_cycle * 1e-3)}
    # ---- the trainer's loop: varying-T batches through the engine (buckets of 8 phonemes / 32 frames), pinned-memory feeder
    mt = FastSpeech2(c1.preprocess_config, c1.model_config, 65, device=dev, seed=1234).train()
    ot = ScheduledOptim(mt, c1.train_config, c1.model_config, 0)
    eng = TrainEngine(mt, ot, c1, loss_fn)
    mi = cfg.get("mi355x", {})
    bucket = (int(mi.get("l_bucket", 8)), int(mi.get("t_bucket", 32)), int(cfg.model_config["max_seq_len"]))
    host = [tuple(x.numpy() if torch.is_tensor(x) else x for x in make_batch(B, L - (i % 3), seed=2000 + i)) for i in range(12)]
    frames = [int(np.asarray(b[7]).sum()) for b in host]
    step = [0]

    def run(batches):
        last = None
        for b in DeviceFeeder(batches, dev, bucket=bucket):
            step[0] += 1
            last, _ = eng.step(b, step[0])
        return last
    run(host); run(host)                       # first sight (eager) and second sight (capture) of every shape
    torch.cuda.synchronize()
    n_loops = max(1, (max(steps, 50) + len(host) - 1) // len(host))
    t0 = time.perf_counter()
    for _ in range(n_loops):
        last = run(host)
    last.cpu()
    dt = time.perf_counter() - t0
    n = n_loops * len(host)
    out["ms_per_step_train_loop"] = 1e3 * dt / n
    out["train_loop"] = {"steps": n, "distinct_batches": len(host), "graphs": len(eng._graphs), "engine": dict(eng.stats),
                         "T_max_range": [int(min(int(b[8]) for b in host)), int(max(int(b[8]) for b in host))],
                         "valid_mel_frames_per_s": n_loops * sum(frames) / dt,
                         "what": "TrainEngine.step over DeviceFeeder (host-side bucket padding, pinned H2D one batch ahead), hipGraph replay per shape"}
    return out


def dp1_leg_isolated(args):
    """`dp1_leg` in a child process, started before this process initialises the GPU.  The leg creates an RCCL communicator and
    captures collectives into a hipGraph; ProcessGroupNCCL's watchdog thread aborted the whole process once while doing that (a HIP
    event query during the capture, see tts_king_amd/graph.py), and an abort here would cost the bench line.  Under rocprofv3 the
    profiler's preload has already initialised the GPU in this process, and starting another program from such a process is not
    allowed on this pool: the leg is skipped there."""
    import subprocess
    if "rocprof" in os.environ.get("LD_PRELOAD", "") or any(k.startswith("ROCPROF") for k in os.environ):
        return {"skipped": "running under rocprofv3: no child process may be started"}
    cmd = [sys.executable, os.path.abspath(__file__), "--dp1-child", "--batch", str(args.batch), "--phonemes", str(args.phonemes)]
    try:
        p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
        lines = [ln for ln in p.stdout.decode(errors="replace").splitlines() if ln.startswith("{")]
        if p.returncode == 0 and lines:
            return json.loads(lines[-1])
        return {"error": "child exited with %d" % p.returncode, "stderr_tail": p.stderr.decode(errors="replace")[-400:]}
    except (subprocess.TimeoutExpired, OSError, ValueError) as e:
        return {"error": "%s: %s" % (type(e).__name__, e)}


def dp1_leg(cfg, dev, B, L, steps=30):
    """The data-parallel schedules (FastSpeech2.dp_schedule: "side" = the default, "late") on ONE GPU: the full
    step with GradReducer issuing its bucketed all-reduces over RCCL at world
    size 1 (a collective per gradient bucket on RCCL's stream, backward_native flushing its deferred weight-gradient work whenever
    a bucket completes) — captured in a hipGraph and eager — beside the plain step, and how many grouped-GEMM / reducer / column-sum
    flush launches the DP schedule issues per step."""
    import copy
    import socket
    from tts_king_amd import ops
    from tts_king_amd.fastspeech2 import FastSpeech2
    from tts_king_amd.graph import GraphedTrainStep, make_enqueue
    from tts_king_amd.loss import FastSpeech2Loss
    from tts_king_amd.optimizer import ScheduledOptim
    from tts_king_amd.parallel import GradReducer
    from tts_king_amd.synthetic import make_batch
    from tts_king_amd.train_step import to_device
    own_pg = not dist.is_initialized()
    if own_pg:
        s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
        dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%d" % port, rank=0, world_size=1)
    try:
        c1 = copy.deepcopy(cfg)
        c1.train_config["optimizer"]["grad_acc_step"] = 1
        loss_fn = FastSpeech2Loss(c1.preprocess_config, c1.model_config)
        batch = to_device(make_batch(B, L, seed=1234), dev)
        out = None
        default = switches.get("TTSK_DP_SCHEDULE")
        for sched in [default] + [x for x in ("side", "late") if x != default]:
            model = FastSpeech2(c1.preprocess_config, c1.model_config, 65, device=dev, seed=1234).train()
            model.dp_schedule = sched
            opt = ScheduledOptim(model, c1.train_config, c1.model_config, 0)
            buckets = model.grad_buckets(cfg.mi355x.dp_bucket_mb)
            red = GradReducer(model.flat_buffers()[1], buckets, model.group_offsets(), force_collectives=True)
            enq = make_enqueue(model, opt, c1, loss_fn, reducer=red, grad_scale=1.0)
            counts = {}
            for _ in range(2):
                enq(batch)
            torch.cuda.synchronize()
            ops.LAUNCH_COUNTS = counts
            enq(batch)
            torch.cuda.synchronize()
            ops.LAUNCH_COUNTS = None
            rec = {"schedule": sched, "buckets": len(buckets), "bucket_mb": cfg.mi355x.dp_bucket_mb, "launches_per_step": dict(counts),
                   "dp1_reducer_ms_per_step_eager": _time_loop(lambda: enq(batch), 10)}
            g = None
            try:
                g = GraphedTrainStep(enq, batch, warmup=0)
                for _ in range(3):
                    g.run()
                rec["dp1_reducer_ms_per_step"] = _time_loop(lambda: g.run(), steps)
            except Exception as e:          # RCCL refused the capture: the eager number stands
                rec["dp1_reducer_ms_per_step"] = None
                rec["graph_capture_error"] = str(e)[:200]
            if out is None:
                out = rec
                out["by_schedule"] = {}
            out["by_schedule"][sched] = {k: rec[k] for k in ("dp1_reducer_ms_per_step", "dp1_reducer_ms_per_step_eager", "launches_per_step")}
            del g, enq, red, opt, model
        return out
    finally:
        if own_pg:
            dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--phonemes", type=int, default=64)
    ap.add_argument("--no-graph", action="store_true", help="launch eagerly instead of replaying a hipGraph")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-hifi", action="store_true")
    ap.add_argument("--no-mel", action="store_true")
    ap.add_argument("--no-e2e", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the eager / grad_acc_step=4 / trainer-loop / DP-schedule legs")
    ap.add_argument("--dp1-child", action="store_true", help=argparse.SUPPRESS)
    args = ap.parse_args()
    from tts_king_amd.hostcpu import fit_torch_threads
    fit_torch_threads()           # the host-side legs (eager launches, the trainer's loop) on the cores this process is granted
    if args.dp1_child:            # the DP-schedule leg in a process of its own (see dp1_leg_isolated)
        real_stdout = os.dup(1)
        os.dup2(2, 1)
        from tts_king_amd.config import default_config
        torch.cuda.set_device(0)
        rec = dp1_leg(default_config(), "cuda:0", args.batch, args.phonemes)
        os.write(real_stdout, (json.dumps(rec) + "\n").encode())
        return
    from tts_king_amd import launch
    if launch.wants_spawn(args.gpus):
        # `python bench.py --gpus N` without torch.distributed.run: this process starts the N ranks itself (before any GPU call —
        # it never makes one), relays rank 0's JSON line and exits non-zero if a rank failed or fewer than N devices are visible
        raise SystemExit(launch.spawn_ranks(args.gpus, [sys.executable, os.path.abspath(__file__)] + sys.argv[1:]))
    dp1_rec = None
    if args.gpus == 1 and int(os.environ.get("WORLD_SIZE", "1")) == 1 and not args.no_extra:
        dp1_rec = dp1_leg_isolated(args)          # before this process touches the GPU
    # ONE JSON line on stdout, nothing else: RCCL prints a version banner to the C-level stdout when a communicator is created
    # (and flushes it at exit, i.e. AFTER the JSON line).  Everything the process or its libraries print goes to stderr; the
    # record is written to the original stdout descriptor at the very end.
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)

    from tts_king_amd.config import default_config
    from tts_king_amd.fastspeech2 import FastSpeech2
    from tts_king_amd.graph import GraphedTrainStep, make_enqueue
    from tts_king_amd.loss import FastSpeech2Loss
    from tts_king_amd.optimizer import ScheduledOptim
    from tts_king_amd.parallel import GradReducer, init_distributed
    from tts_king_amd.synthetic import make_batch
    from tts_king_amd.train_step import to_device

    rank, world, local = init_distributed()
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if torch.cuda.device_count() <= local:
        raise SystemExit("rank %d wants cuda:%d but %d HIP device(s) are visible" % (rank, local, torch.cuda.device_count()))
    dev = "cuda:%d" % local
    torch.cuda.set_device(local)
    cfg = default_config()
    cfg.train_config["optimizer"]["grad_acc_step"] = 1
    B, L = args.batch, args.phonemes
    model = FastSpeech2(cfg.preprocess_config, cfg.model_config, 65, device=dev, seed=1234).train()
    opt = ScheduledOptim(model, cfg.train_config, cfg.model_config, 0)
    loss_fn = FastSpeech2Loss(cfg.preprocess_config, cfg.model_config)
    cpu_batch = make_batch(B, L, seed=1234 + rank)
    batch = to_device(cpu_batch, dev)
    T = int(batch[8])
    frames = int(cpu_batch[7].sum())

    reducer = None
    if world > 1:
        opt.state[2] += rank        # same weights on every rank, a dropout key of its own per rank (as train.py)
        reducer = GradReducer(model.flat_buffers()[1], model.grad_buckets(cfg.mi355x.dp_bucket_mb), model.group_offsets())
    enqueue = make_enqueue(model, opt, cfg, loss_fn, reducer=reducer,
                           grad_scale=reducer.grad_scale(1) if reducer else None)
    use_graph = not args.no_graph and (world == 1 or switches.get("TTSK_DP_GRAPH") != "0")
    step = lambda: enqueue(batch)
    if use_graph:
        # the data-parallel step (RCCL bucket all-reduces included) is captured as well; two eager steps first so that
        # communicators, lazy allocations and the split-K plans exist before the capture
        try:
            if world > 1:
                for _ in range(2):
                    enqueue(batch)
                torch.cuda.synchronize()
                dist.barrier()
            g = GraphedTrainStep(enqueue, batch, warmup=0 if world > 1 else 2)
            step = lambda: g.run()
        except Exception as e:      # capture refused: launch eagerly (still correct, host-bound)
            if rank == 0:
                print("graph capture failed (%s): eager launches" % e, file=sys.stderr)
            use_graph = False
            # a capture that died half way leaves host-side bookkeeping advanced: put it back (as TrainEngine does)
            if reducer is not None:
                reducer.reset()
            model.grads_partial = False
            model.abort_step()
            torch.cuda.synchronize()

    # A freshly captured graph is not at its steady state: its first replays run 2-6 % slow (tools/debug/fs2_warm_curve.py: 2.556, 2.460, 2.431,
    # 2.414 ms in blocks of five, 2.41 from the 16th on — new buffers, cold translations, a clock that has not settled).  A training run replays
    # the graph millions of times, so the timed region starts behind at least SETTLE untimed replays: the W warmup steps asked for, preceded by
    # what a small W leaves missing.  Reported as config.settle_replays.
    SETTLE = 20
    settle = max(0, SETTLE - args.warmup) if use_graph else 0
    for _ in range(settle):
        out = step()
    for _ in range(args.warmup):
        out = step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    stats = torch.tensor([dt, float(frames)], dtype=torch.float64, device=dev)
    if world > 1:
        tmax = stats[0:1].clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        tot = stats[1:2].clone()
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
        dt, total_frames = float(tmax), float(tot)
    else:
        total_frames = float(frames)
    losses = out[0].cpu().tolist()
    ms = 1e3 * dt / args.steps
    value = total_frames * args.steps / dt

    if rank == 0:
        flops = fs2_flops_per_step(B, L, T)
        rec = {
            "metric": "FS2 train mel-frames/sec/GPU + HiFi-GAN RTF (22.05 kHz), batch=16",
            "value": value, "unit": "valid mel-frames/s (whole job)", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16",
            "data": "synthetic",
            "config": {"workload": "FS2 train step bf16 on 1xMI355X per rank, 65-speaker embedding, batch=16, 80-bin mel, "
                                   "256-d FFT blocks (BASELINE.json configs[1]); full step = fwd+loss+bwd+clip+Adam, dropout on",
                       "global_batch": B * world, "batch_per_gpu": B, "phonemes": L, "T_max": T, "valid_frames_per_gpu": frames,
                       "padded_frames_per_gpu": B * T, "grad_acc_step": 1, "parallelism": "dp%d" % world, "settle_replays": settle,
                       "untimed_replays_before_timing": {"fs2": settle + args.warmup, "hifi_gan": 20},
                       "launch": ("hipGraph replay" + (" incl. RCCL bucketed all-reduce" if world > 1 else "")) if use_graph
                                 else "eager (+RCCL bucketed all-reduce on a side stream)",
                       # what the collective layer itself saw (a SCALE record can be checked for "RCCL ran with N ranks"): the size of
                       # the process group, its backend, and the bytes of every bucket's all-reduce in launch order, per step
                       "collective_world_size": dist.get_world_size() if dist.is_initialized() else 1,
                       "collective_backend": dist.get_backend() if dist.is_initialized() else None,
                       "allreduce_bucket_bytes": reducer.bucket_bytes() if reducer is not None else [],
                       "allreduces_per_step": (len(reducer.history[-1]) if (reducer is not None and reducer.history) else 0),
                       "collective_timeout_s": (__import__("tts_king_amd.parallel", fromlist=["dist_timeout_s"]).dist_timeout_s() if world > 1 else None)},
            "mel_frames_per_s_per_gpu": value / world,
            "model_tflops": flops * world / (ms * 1e-3) / 1e12,
            "step_mfma_roofline_frac": flops / (ms * 1e-3) / 1e12 / PEAK_MFMA_BF16_TFLOPS,
            "final_losses": {"total": losses[0], "mel": losses[1], "pitch": losses[2], "energy": losses[3], "duration": losses[4]},
        }
        if switches.unknown_in_environment():
            rec["unknown_switches"] = switches.unknown_in_environment()      # TTSK_* variables nothing reads (tts_king_amd/switches.py)
        # the metric's second half right behind its first: both headline legs see the chip as the W warmup + K timed steps left it, neither
        # the state ~30 s of auxiliary legs (eager, grad_acc, trainer loop, roofline brackets) leave behind (HiFi-GAN measured 2 % slower there)
        if world == 1 and not args.no_hifi:
            try:
                from tts_king_amd.hifi_bench import hifi_rtf
                rec["hifi_gan"] = hifi_rtf(cfg, dev)
            except ImportError:
                rec["hifi_gan"] = None
        if not args.no_roofline:
            eager = make_enqueue(model, opt, cfg, loss_fn, reducer=None)
            rec["roofline"] = step_roofline(eager, batch)
            # the family with the most FLOPs behind one symbol's worth of launches (round 3's headline kernel), kept for continuity:
            # dwconv runs as two launches per step — six decoder weights = 192 workgroups (one per CU) beside the encoder-side dX
            # chain, four encoder weights = 128 workgroups in the final phase; FLOPs counted for all B*T rows, PAD rows are skipped
            dwc = [r for r in rec["roofline"]["families"] if r["family"] == "dwconv"]
            if dwc:
                rec["roofline"]["dwconv_grid"] = {"decoder_launch_workgroups": 192, "encoder_launch_workgroups": 128, "cus": 256,
                                                  "frac_of_peak": dwc[0]["frac_of_peak"]}
        if world == 1 and not args.no_extra:
            rec.update(extra_train_legs(cfg, dev, B, L, steps=args.steps))
            rec["dp_schedule_1gpu"] = dp1_rec
        if world == 1 and not args.no_e2e:
            rec["e2e_synth"] = e2e_synth_leg(cfg, dev, with_cpu=not args.no_cpu_baseline)
        if world == 1 and not args.no_mel:
            rec["mel_extraction"] = mel_extraction_leg(cfg, dev, with_cpu=not args.no_cpu_baseline)
        if world == 1 and not args.no_cpu_baseline:
            rec["cpu_baseline"] = cpu_baseline(cfg, B, L)
            if rec.get("hifi_gan"):
                rec["hifi_gan"]["cpu_baseline"] = hifi_cpu_baseline(cfg)
        sys.stdout.flush()
        full = write_full_record(rec)
        sys.stderr.write("bench.py full record (%s):\n%s\n" % (full, json.dumps(rec)))
        sys.stderr.flush()
        os.write(real_stdout, (compact_record(rec, full) + "\n").encode())
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
