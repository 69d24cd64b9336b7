#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the REFERENCE implementation (build container only).

The reference (/root/reference, pure Python/PyTorch) is imported with three empty stub modules for
preprocessing-only dependencies that are absent here (pycwt, unidecode, inflect — none is executed on
the hot path), its weights are filled with `tts_king_amd.synthetic.seeded_fill`, and inputs/outputs of
the hot path are stored as small fixtures.  Nothing from the reference travels: the fixtures are data.

    python tools/make_goldens.py            # rewrites tests/golden/
"""
import os
import sys
import types

import numpy as np
import torch
import yaml

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
sys.path.insert(0, REPO)
sys.path.insert(0, REF)
for name in ("pycwt", "unidecode", "inflect"):
    sys.modules[name] = types.ModuleType(name)
sys.modules["unidecode"].unidecode = lambda s: s
sys.modules["inflect"].engine = lambda: None

from tts_king_amd.synthetic import seeded_fill, make_batch, make_mel  # noqa: E402


class AD(dict):
    def __init__(self, d):
        super().__init__({k: AD(v) if isinstance(v, dict) else v for k, v in d.items()})
    __getattr__ = dict.__getitem__


cfg = AD(yaml.safe_load(open(os.path.join(REF, "config.yaml"))))
cfg.preprocess_config.path["preprocessed_path"] = os.path.join(REF, "pretrained")

from fs_two.model import FastSpeech2, FastSpeech2Loss, ScheduledOptim  # noqa: E402
from fs_two.model.modules import LengthRegulator  # noqa: E402
from hifi.models import Generator  # noqa: E402

OUT = os.path.join(REPO, "tests", "golden")
os.makedirs(OUT, exist_ok=True)
N_SPK = 65
WEIGHT_SEED = 7


def npy(t):
    return t.detach().cpu().numpy() if torch.is_tensor(t) else np.asarray(t)


def new_model():
    torch.manual_seed(0)
    m = FastSpeech2(cfg.preprocess_config, cfg.model_config, N_SPK)
    seeded_fill(m.state_dict(), WEIGHT_SEED)
    return m


def small_grads(model, keys):
    return {("grad/" + k): npy(dict(model.named_parameters())[k].grad) for k in keys}


SMALL = ["mel_linear.bias", "encoder.layer_stack.0.slf_attn.layer_norm.weight",
         "decoder.layer_stack.5.pos_ffn.w_2.bias", "variance_adaptor.duration_predictor.linear_layer.weight",
         "postnet.convolutions.4.1.weight", "decoder.layer_stack.0.slf_attn.w_qs.bias",
         "variance_adaptor.energy_predictor.conv_layer.layer_norm_1.bias"]


def g1_eval_teacher_forced():
    m = new_model().eval()
    b = make_batch(2, 64, seed=11, ragged=True, n_speakers=N_SPK)
    with torch.no_grad():
        o = m(*b[2:])
    np.savez_compressed(os.path.join(OUT, "fs2_eval_tf.npz"), B=2, L=64, seed=11, weight_seed=WEIGHT_SEED,
                        mel=npy(o[0]), pitch=npy(o[1]), energy=npy(o[2]), logd=npy(o[3]),
                        mel_lens=npy(o[8]), post=npy(o[9]))
    print("G1 mel", tuple(o[0].shape), "absmax", float(o[0].abs().max()))


DUR_BIAS = 1.3   # random weights predict log-durations near 0 (=> almost no frames); shift them to ~e^1.3-1


def g2_eval_free_running():
    m = new_model().eval()
    with torch.no_grad():
        m.variance_adaptor.duration_predictor.linear_layer.bias.fill_(DUR_BIAS)
    b = make_batch(2, 64, seed=12, ragged=True, n_speakers=N_SPK)
    with torch.no_grad():
        o = m(b[2], b[3], b[4], b[5], d_control=0.9, p_control=1.5, e_control=1.2)
    np.savez_compressed(os.path.join(OUT, "fs2_eval_free.npz"), B=2, L=64, seed=12, weight_seed=WEIGHT_SEED,
                        controls=np.array([0.9, 1.5, 1.2]), dur_bias=DUR_BIAS, d_rounded=npy(o[4]), mel_lens=npy(o[8]),
                        mel=npy(o[0]), post=npy(o[9]), pitch=npy(o[1]), energy=npy(o[2]), logd=npy(o[3]))
    print("G2 mel", tuple(o[0].shape), "mel_lens", o[8].tolist())


def _no_dropout():
    import torch.nn.functional as F
    orig = F.dropout
    F.dropout = lambda x, p=0.5, training=True, inplace=False: x
    return orig


def g3_train_no_dropout():
    import torch.nn.functional as F
    orig = _no_dropout()
    try:
        m = new_model().train()
        loss_fn = FastSpeech2Loss(cfg.preprocess_config, cfg.model_config)
        b = make_batch(2, 64, seed=13, ragged=True, n_speakers=N_SPK)
        o = m(*b[2:])
        ls = loss_fn(b, o)
        ls[0].backward()
    finally:
        F.dropout = orig
    named = dict(m.named_parameters())
    keys = sorted(k for k, p in named.items() if p.grad is not None)
    none_keys = sorted(k for k, p in named.items() if p.grad is None and p.requires_grad)
    gn = np.array([float(named[k].grad.norm()) for k in keys])
    sdm = m.state_dict()
    bn = {("bn/" + k): npy(v) for k, v in sdm.items() if "running_" in k and ".4.1." in k}
    np.savez_compressed(os.path.join(OUT, "fs2_train_p0.npz"), B=2, L=64, seed=13, weight_seed=WEIGHT_SEED,
                        losses=np.array([float(l.sum()) for l in ls]), grad_keys=np.array(keys),
                        grad_norms=gn, none_keys=np.array(none_keys), mel=npy(o[0]), post=npy(o[9]),
                        **small_grads(m, SMALL), **bn)
    print("G3 losses", [round(float(l.sum()), 4) for l in ls], "n_grads", len(keys), "none", len(none_keys))


def g4_length_regulator():
    lr = LengthRegulator()
    g = torch.Generator().manual_seed(5)
    x = torch.randn(3, 7, 4, generator=g)
    d = torch.tensor([[1, 0, 3.6, 2, -1, 0, 1.2], [2, 2, 0, 0, 1, 0.9, 1.0], [0, 0, 0, 0, 0, 0, 0]])
    out = {}
    o, ml = lr(x, d, None)
    out.update(x=npy(x), d=npy(d), out_none=npy(o), len_none=npy(ml))
    o, ml = lr(x, d, 4)            # crop: output (3,4,4) while mel_len keeps the uncropped totals
    out.update(out_crop4=npy(o), len_crop4=npy(ml))
    o, ml = lr(x, d, 12)           # pad beyond the batch max
    out.update(out_pad12=npy(o), len_pad12=npy(ml))
    di = torch.randint(0, 9, (4, 33), generator=g)
    xi = torch.randn(4, 33, 8, generator=g)
    o, ml = lr(xi, di, int(di.sum(1).max()))
    out.update(xi=npy(xi), di=npy(di), out_int=npy(o), len_int=npy(ml))
    np.savez_compressed(os.path.join(OUT, "length_regulator.npz"), **out)
    print("G4 len_none", out["len_none"].tolist(), "crop", out["out_crop4"].shape)


def g6_adam_steps():
    import torch.nn.functional as F
    import torch.nn as nn
    res = {}
    for s in (1, 4000, 300001):
        orig = _no_dropout()
        try:
            m = new_model().train()
            opt = ScheduledOptim(m, cfg.train_config, cfg.model_config, s - 1)
            loss_fn = FastSpeech2Loss(cfg.preprocess_config, cfg.model_config)
            b = make_batch(2, 64, seed=14, ragged=True, n_speakers=N_SPK)
            before = {k: v.clone() for k, v in m.state_dict().items()}
            o = m(*b[2:])
            ls = loss_fn(b, o)
            (ls[0] / 1).backward()
            gnorm = nn.utils.clip_grad_norm_(m.parameters(), cfg.train_config["optimizer"]["grad_clip_thresh"])
            opt.step_and_update_lr()
            opt.zero_grad()
        finally:
            F.dropout = orig
        after = m.state_dict()
        res["lr_%d" % s] = opt._optimizer.param_groups[0]["lr"]
        res["gnorm_%d" % s] = float(gnorm)
        for k in SMALL:
            res["delta_%d/%s" % (s, k)] = npy(after[k] - before[k])
        res["delta_norm_%d" % s] = np.array([float((after[k] - before[k]).float().norm()) for k in sorted(after)])
    res["keys"] = np.array(sorted(after))
    np.savez_compressed(os.path.join(OUT, "adam_steps.npz"), B=2, L=64, seed=14, weight_seed=WEIGHT_SEED, **res)
    print("G6 lr", [res["lr_%d" % s] for s in (1, 4000, 300001)], "gnorm", res["gnorm_1"])


def g7_hifigan():
    torch.manual_seed(0)
    gen = Generator(cfg.hifi)
    sd_wn = gen.state_dict()                       # weight-normed layout: *.weight_g / *.weight_v
    seeded_fill(sd_wn, WEIGHT_SEED)
    wn_keys = sorted(sd_wn.keys())
    gen.remove_weight_norm()
    gen.eval()
    folded = gen.state_dict()
    mel = make_mel(2, 32, seed=21)
    with torch.no_grad():
        wav = gen(mel)
        i16 = (wav * cfg.hifi.MAX_WAV_VALUE).cpu().numpy().astype("int16")
    probe = ["ups.0.weight", "ups.3.weight", "conv_pre.weight", "resblocks.11.convs1.2.weight", "conv_post.weight"]
    np.savez_compressed(os.path.join(OUT, "hifi_b2_t32.npz"), B=2, T=32, seed=21, weight_seed=WEIGHT_SEED,
                        wav=npy(wav), int16=i16, n_wn_keys=len(wn_keys), n_folded_keys=len(folded),
                        **{("fold/" + k): npy(folded[k]).ravel()[:64] for k in probe},
                        **{("foldnorm/" + k): float(folded[k].norm()) for k in probe})
    print("G7 wav", tuple(wav.shape), "absmax", float(wav.abs().max()), "rms", float(wav.pow(2).mean().sqrt()),
          "keys", len(wn_keys), len(folded))


def g8_shapes():
    m = new_model()
    sd = m.state_dict()
    np.savez_compressed(os.path.join(OUT, "fs2_state_dict_spec.npz"), keys=np.array(list(sd.keys())),
                        shapes=np.array([";".join(map(str, v.shape)) for v in sd.values()]),
                        dtypes=np.array([str(v.dtype) for v in sd.values()]),
                        trainable=np.array([k for k, p in m.named_parameters() if p.requires_grad]),
                        n_params=sum(p.numel() for p in m.parameters()))
    torch.manual_seed(0)
    g = Generator(cfg.hifi)
    sdw = g.state_dict()
    g.remove_weight_norm()
    sdf = g.state_dict()
    np.savez_compressed(os.path.join(OUT, "hifi_state_dict_spec.npz"), wn_keys=np.array(list(sdw.keys())),
                        wn_shapes=np.array([";".join(map(str, v.shape)) for v in sdw.values()]),
                        keys=np.array(list(sdf.keys())),
                        shapes=np.array([";".join(map(str, v.shape)) for v in sdf.values()]))
    print("G8 fs2 keys", len(sd), "hifi", len(sdw), len(sdf))


def g9_text():
    """Symbol inventory (data asset pretrained/symbols.json: the id of a symbol is its position + the model's vocabulary
    is len + 1, Models.py:40) and known-answer vectors of `text_to_sequence` (examples.ipynb cell 2 plus a few more)."""
    import json
    from fs_two.text import text_to_sequence
    from fs_two.text.symbols import symbols
    with open(os.path.join(REPO, "pretrained", "symbols.json"), "w") as f:
        json.dump(list(symbols), f, ensure_ascii=False)
    cases = ["{R A B O0 T A T0 I R A B O0 T A T0 sp S K A Z A0 L O0 N sp}",
             "{P R I0 V E0 T sp M I0 R sp}",
             "Turn left on {HH AW1 S S T AH0 N} Street.",
             "{sil} {spn} a-b, c!",
             "{NOSUCH R A} ~_x"]
    out = {c: text_to_sequence(c, []) for c in cases}
    with open(os.path.join(OUT, "text_to_sequence.json"), "w") as f:
        json.dump({"n_symbols": len(symbols), "cases": out}, f, ensure_ascii=False, indent=1)
    print("G9 symbols", len(symbols), {k: len(v) for k, v in out.items()})


def synthetic_samples(n, seed):
    """Seeded per-utterance sample dicts with the fields Dataset.__getitem__ returns (fs_two/dataset.py:117-131)."""
    rng = np.random.RandomState(seed)
    out = []
    for i in range(n):
        L = int(rng.randint(5, 40))
        dur = rng.randint(1, 6, size=L)
        T = int(dur.sum())
        out.append({"id": "utt%03d" % i, "speaker": int(rng.randint(0, 65)), "text": rng.randint(1, 207, size=L),
                    "raw_text": "raw %d" % i, "mel": rng.randn(T, 80).astype(np.float32), "energy": rng.randn(L).astype(np.float32),
                    "duration": dur, "pitch_raw": rng.randn(L).astype(np.float32), "pitch_mean": np.float32(rng.randn()),
                    "pitch_std": np.float32(abs(rng.randn()) + 0.1), "pitch_cwt": rng.randn(L, 11).astype(np.float32)})
    return out


def g10_collate():
    """Reference collate (sort by phoneme count, cut into batches, pad) on seeded samples: fs_two/dataset.py:158-225."""
    from fs_two.dataset import Dataset as RefDataset
    res = {}
    for tag, (n, bs, sort, drop) in {"a": (11, 4, True, True), "b": (11, 4, True, False), "c": (8, 4, False, True)}.items():
        dummy = types.SimpleNamespace(sort=sort, batch_size=bs, drop_last=drop)
        dummy.reprocess = lambda data, idxs: RefDataset.reprocess(dummy, data, idxs)
        batches = RefDataset.collate_fn(dummy, synthetic_samples(n, 77))
        res[tag + "/n"] = len(batches)
        for bi, b in enumerate(batches):
            res["%s/%d/ids" % (tag, bi)] = np.array(b[0])
            for fi in (2, 3, 4, 6, 7, 9, 10, 11, 12, 13, 14):
                res["%s/%d/%d" % (tag, bi, fi)] = np.asarray(b[fi])
            res["%s/%d/max" % (tag, bi)] = np.array([b[5], b[8]])
    np.savez_compressed(os.path.join(OUT, "collate.npz"), **res)
    print("G10 collate", {k: int(v) for k, v in res.items() if k.endswith("/n")})


def mel_test_signal(Bsz=2, n=8192, seed=5):
    """Seeded test waveform: a few partials, a chirp, noise, and a 40 dB quieter second half (values in [-1, 1])."""
    g = torch.Generator().manual_seed(seed)
    t = torch.arange(n, dtype=torch.float64) / 22050.0
    rows = []
    for b in range(Bsz):
        y = 0.3 * torch.sin(2 * np.pi * (220.0 * (b + 1)) * t) + 0.2 * torch.sin(2 * np.pi * 3100.0 * t + 0.5)
        y = y + 0.2 * torch.sin(2 * np.pi * (500.0 + 9000.0 * t) * t) + 0.05 * (torch.rand(n, generator=g, dtype=torch.float64) * 2 - 1)
        env = torch.ones(n, dtype=torch.float64)
        env[n // 2:] = 0.01
        rows.append((y * env).clamp(-1, 1))
    return torch.stack(rows).float()


def g11_mel():
    """Mel extraction.  Neither reference function runs here as written (hifi/meldataset.py:64 calls
    torch.stft without `return_complex`, which current torch rejects; fs_two/audio/stft.py:77 hard-codes .cuda(3)), and
    librosa is absent.  The golden therefore comes from torch.stft itself with the reference's arguments
    (meldataset.py:57-66) + the librosa filterbank as restated in oracle/audio.py (pinned there by librosa's docstring
    known answers), and from F.conv1d with the reference's windowed Fourier basis for the TacotronSTFT variant."""
    from oracle import audio as OA
    p = cfg.preprocess_config.preprocessing
    n_fft, hop, win = p.stft.filter_length, p.stft.hop_length, p.stft.win_length
    n_mel, sr, fmin, fmax = p.mel.n_mel_channels, p.audio.sampling_rate, p.mel.mel_fmin, p.mel.mel_fmax
    y = mel_test_signal()
    basis = torch.from_numpy(OA.mel_filterbank(sr, n_fft, n_mel, fmin, fmax))
    pad = int((n_fft - hop) / 2)
    yp = torch.nn.functional.pad(y.unsqueeze(1), (pad, pad), mode="reflect").squeeze(1)
    spec = torch.stft(yp, n_fft, hop_length=hop, win_length=win, window=torch.hann_window(win), center=False,
                      pad_mode="reflect", normalized=False, onesided=True, return_complex=True)
    spec = torch.sqrt(torch.view_as_real(spec).pow(2).sum(-1) + 1e-9)
    mel_hifi = torch.log(torch.clamp(torch.matmul(basis, spec), min=1e-5))
    mel_taco, energy = OA.tacotron_mel(y, n_fft, hop, win, n_mel, sr, fmin, fmax)
    np.savez_compressed(os.path.join(OUT, "mel_extraction.npz"), y=npy(y), mel_hifi=npy(mel_hifi), mel_taco=npy(mel_taco),
                        energy=npy(energy), params=np.array([n_fft, hop, win, n_mel, sr, fmin, fmax], dtype=np.float64))
    print("G11 mel", tuple(mel_hifi.shape), tuple(mel_taco.shape), float(mel_hifi.min()), float(mel_hifi.max()))


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "mel":
        g11_mel()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "text":
        g9_text()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "collate":
        g10_collate()
        sys.exit(0)
    g1_eval_teacher_forced()
    g2_eval_free_running()
    g3_train_no_dropout()
    g4_length_regulator()
    g6_adam_steps()
    g7_hifigan()
    g8_shapes()
    g9_text()
    g10_collate()
    g11_mel()
