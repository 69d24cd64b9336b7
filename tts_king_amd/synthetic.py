"""Synthetic inputs and deterministic weight fills.

No pretrained weights and no dataset ship with the reference (README.md:38-39), so every parity
test and every benchmark runs on

* `seeded_fill(state_dict, seed)`: a weight fill that does not depend on module construction order
  (keys are visited in sorted order, one `torch.Generator` per key), and
* `make_batch(B, L, seed)`: the 15-tuple batch layout `Dataset.reprocess` emits
  (reference: fs_two/dataset.py:188-204), filled with seeded random phonemes / durations / mels.

Both are pure CPU torch so the same numbers come out in the build container and on the GPU box.
"""
import math
import re

import torch

N_VOCAB = 207          # len(symbols) + 1  (reference: fs_two/transformer/Models.py:40)
KEEP_AS_BUILT = ("position_enc", "pitch_bins", "energy_bins", "num_batches_tracked")


def _fan_in(shape):
    if len(shape) <= 1:
        return 1
    n = 1
    for s in shape[1:]:
        n *= s
    return n


def seeded_fill(sd, seed=0, conv_transpose_keys=()):
    """In-place deterministic fill of a state_dict-like mapping of tensors.

    LayerNorm/BatchNorm weights ~ 1 + 0.1 N(0,1); running_var = 0.5 + U(0,1); biases and
    running_mean 0.02 N(0,1); matrices N(0,1)/sqrt(fan_in); weight_g (weight-norm gains) 0.4 + 0.6 U(0,1).
    `position_enc`, `*_bins` and `num_batches_tracked` are left as constructed.
    """
    for idx, key in enumerate(sorted(sd.keys())):
        t = sd[key]
        if any(k in key for k in KEEP_AS_BUILT):
            continue
        g = torch.Generator().manual_seed(seed + idx)
        shape = tuple(t.shape)
        leaf = key.rsplit(".", 1)[-1]
        is_norm = ("layer_norm" in key) or bool(re.search(r"postnet\.convolutions\.\d+\.1\.", key)) or (".net.2." in key)
        if leaf == "running_var":
            v = 0.5 + torch.rand(shape, generator=g)
        elif leaf == "running_mean":
            v = 0.02 * torch.randn(shape, generator=g)
        elif leaf == "weight_g":
            v = 0.4 + 0.6 * torch.rand(shape, generator=g)
        elif leaf == "bias":
            v = 0.02 * torch.randn(shape, generator=g)
        elif leaf == "weight" and is_norm:
            v = 1.0 + 0.1 * torch.randn(shape, generator=g)
        else:
            fan = _fan_in(shape)
            if key in conv_transpose_keys or (key.startswith("ups.") and len(shape) == 3):
                fan = shape[0] * shape[2] / 4.0  # ConvTranspose1d: (Cin, Cout, k); ~k/stride taps hit an output
            v = torch.randn(shape, generator=g) / math.sqrt(max(fan, 1))
        with torch.no_grad():
            t.copy_(v.to(t.dtype))
    return sd


def make_batch(B=16, L=64, seed=1234, ragged=False, n_speakers=65, n_mel=80, dur_hi=12):
    """Canonical synthetic batch, CPU tensors, reference dtypes.

    Returns the 15-tuple (ids, raw_texts, speakers, texts, src_lens, max_src_len, mels, mel_lens,
    max_mel_len, energies, durations, pitches_raw, pitches_cwt, pitches_mean, pitches_std).
    """
    g = torch.Generator().manual_seed(seed)
    if ragged:
        src = torch.randint(L // 2, L + 1, (B,), generator=g)
        src[0] = L
    else:
        src = torch.full((B,), L, dtype=torch.int64)
    texts = torch.randint(1, N_VOCAB, (B, L), generator=g)
    dur = torch.randint(1, dur_hi, (B, L), generator=g)
    pad = torch.arange(L)[None, :] >= src[:, None]
    texts[pad] = 0
    dur[pad] = 0
    mel_lens = dur.sum(1)
    T = int(mel_lens.max())
    mels = torch.randn(B, T, n_mel, generator=g)
    mels[torch.arange(T)[None, :] >= mel_lens[:, None]] = 0
    pitch = torch.randn(B, L, generator=g)
    energy = torch.randn(B, L, generator=g)
    pitch[pad] = 0
    energy[pad] = 0
    spk = torch.randint(0, n_speakers, (B,), generator=g)
    ids = ["utt%04d" % i for i in range(B)]
    return (ids, ids, spk, texts, src, L, mels, mel_lens, T, energy, dur, pitch,
            torch.zeros(B, L, 11), torch.zeros(B), torch.ones(B))


def make_mel(B=8, T=384, seed=1234, n_mel=80):
    """HiFi-GAN input: randn scaled to the log-mel range."""
    g = torch.Generator().manual_seed(seed)
    return torch.randn(B, n_mel, T, generator=g) * 2.0 - 5.0
