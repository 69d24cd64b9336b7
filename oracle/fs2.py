"""ORACLE (test infrastructure, not product code): CPU fp32 restatement of the reference's
FastSpeech2 train step, written as plain functions over a `state_dict` with the reference's key names.

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import this package.
The product path (`tts_king_amd/`) never does; it fails loudly when the HIP extension is missing.

Pinning: the reference has no tests and no golden vectors for this path, so this
restatement is pinned against outputs of the reference itself, generated in the build container by
`tools/make_goldens.py` (imports /root/reference, writes tests/golden/*.npz) and checked by
`tests/test_oracle_golden.py`.

Every function cites the reference lines it restates.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

PAD = 0  # reference: fs_two/transformer/Constants.py:1


# ----------------------------------------------------------------------------- small pieces

def sinusoid_table(n_position, d_hid):
    """reference: fs_two/transformer/Models.py:10-30 — float64 numpy, then cast to fp32."""
    pos = np.arange(n_position, dtype=np.float64)[:, None]
    j = np.arange(d_hid)[None, :]
    angle = pos / np.power(10000.0, 2.0 * (j // 2) / d_hid)
    tab = np.empty_like(angle)
    tab[:, 0::2] = np.sin(angle[:, 0::2])
    tab[:, 1::2] = np.cos(angle[:, 1::2])
    return torch.from_numpy(tab).float()


def mask_from_lengths(lengths, max_len=None):
    """reference: fs_two/utils/tools.py:121-131 — True marks PAD positions."""
    if max_len is None:
        max_len = int(lengths.max())
    ids = torch.arange(0, max_len, device=lengths.device)[None, :]
    return ids >= lengths[:, None].float()


def _drop(x, p, train):
    return F.dropout(x, p, training=train) if (train and p > 0) else x


# ----------------------------------------------------------------------------- FFT block

def multi_head_attention(sd, pre, x, key_pad, n_head, p, train):
    """reference: fs_two/transformer/SubLayers.py:31-65 and Modules.py:14-24."""
    B, S, D = x.shape
    dk = D // n_head

    def proj(name):
        y = F.linear(x, sd[pre + name + ".weight"], sd[pre + name + ".bias"])
        return y.view(B, S, n_head, dk).permute(2, 0, 1, 3).reshape(n_head * B, S, dk)

    q, k, v = proj("w_qs"), proj("w_ks"), proj("w_vs")
    score = torch.bmm(q, k.transpose(1, 2)) / math.sqrt(dk)            # temperature = sqrt(d_k)
    score = score.masked_fill(key_pad[:, None, :].repeat(n_head, S, 1), float("-inf"))
    attn = torch.softmax(score, dim=2)
    o = torch.bmm(attn, v).view(n_head, B, S, dk).permute(1, 2, 0, 3).reshape(B, S, D)
    o = _drop(F.linear(o, sd[pre + "fc.weight"], sd[pre + "fc.bias"]), p, train)
    return F.layer_norm(o + x, (D,), sd[pre + "layer_norm.weight"], sd[pre + "layer_norm.bias"])


def positionwise_ffn(sd, pre, x, p, train):
    """reference: fs_two/transformer/SubLayers.py:93-101 — Conv1d k9 → ReLU → Conv1d k1 → drop → +x → LN."""
    w1, w2 = sd[pre + "w_1.weight"], sd[pre + "w_2.weight"]
    h = F.relu(F.conv1d(x.transpose(1, 2), w1, sd[pre + "w_1.bias"], padding=(w1.shape[2] - 1) // 2))
    y = F.conv1d(h, w2, sd[pre + "w_2.bias"], padding=(w2.shape[2] - 1) // 2).transpose(1, 2)
    y = _drop(y, p, train)
    return F.layer_norm(y + x, (x.shape[-1],), sd[pre + "layer_norm.weight"], sd[pre + "layer_norm.bias"])


def fft_block(sd, pre, x, pad_mask, n_head, p, train):
    """reference: fs_two/transformer/Layers.py:25-34 — zero PAD rows after each sub-layer."""
    x = multi_head_attention(sd, pre + "slf_attn.", x, pad_mask, n_head, p, train)
    x = x.masked_fill(pad_mask[..., None], 0)
    x = positionwise_ffn(sd, pre + "pos_ffn.", x, p, train)
    return x.masked_fill(pad_mask[..., None], 0)


def _n_layers(sd, pre):
    n = 0
    while (pre + "layer_stack.%d.slf_attn.fc.weight" % n) in sd:
        n += 1
    return n


def encoder(sd, texts, pad_mask, mc, train):
    """reference: fs_two/transformer/Models.py:79-112."""
    L = texts.shape[1]
    d = sd["encoder.src_word_emb.weight"].shape[1]
    emb = F.embedding(texts, sd["encoder.src_word_emb.weight"], padding_idx=PAD)
    if (not train) and L > mc["max_seq_len"]:
        x = emb + sinusoid_table(L, d)[None].to(emb.device)
    else:
        x = emb + sd["encoder.position_enc"][:, :L, :]
    tr = mc["transformer"]
    for i in range(_n_layers(sd, "encoder.")):
        x = fft_block(sd, "encoder.layer_stack.%d." % i, x, pad_mask, tr["encoder_head"],
                      tr["encoder_dropout"], train)
    return x


def decoder(sd, x, pad_mask, mc, train):
    """reference: fs_two/transformer/Models.py:157-189 — truncates to max_seq_len unless eval."""
    T = x.shape[1]
    d = x.shape[2]
    if (not train) and T > mc["max_seq_len"]:
        x = x + sinusoid_table(T, d)[None].to(x.device)
    else:
        T = min(T, mc["max_seq_len"])
        x = x[:, :T, :] + sd["decoder.position_enc"][:, :T, :]
        pad_mask = pad_mask[:, :T]
    tr = mc["transformer"]
    for i in range(_n_layers(sd, "decoder.")):
        x = fft_block(sd, "decoder.layer_stack.%d." % i, x, pad_mask, tr["decoder_head"],
                      tr["decoder_dropout"], train)
    return x, pad_mask


# ----------------------------------------------------------------------------- variance adaptor

def variance_predictor(sd, pre, x, pad_mask, p, train):
    """reference: fs_two/model/modules.py:255-309 (+ Conv :312-355).
    Conv1d k3 → ReLU → LN → Dropout, twice; Linear → squeeze → zero PAD."""
    h = x
    for i in (1, 2):
        w = sd[pre + "conv_layer.conv1d_%d.conv.weight" % i]
        b = sd[pre + "conv_layer.conv1d_%d.conv.bias" % i]
        h = F.relu(F.conv1d(h.transpose(1, 2), w, b, padding=(w.shape[2] - 1) // 2).transpose(1, 2))
        h = F.layer_norm(h, (h.shape[-1],), sd[pre + "conv_layer.layer_norm_%d.weight" % i],
                         sd[pre + "conv_layer.layer_norm_%d.bias" % i])
        h = _drop(h, p, train)
    out = F.linear(h, sd[pre + "linear_layer.weight"], sd[pre + "linear_layer.bias"]).squeeze(-1)
    return out.masked_fill(pad_mask, 0.0)


def length_regulator_index(duration, max_len=None):
    """Index map of reference fs_two/model/modules.py:225-252 + utils/tools.py:369-387.

    `int(d)` truncates toward zero and negatives clamp to 0 (`max(int(d), 0)`, modules.py:244-245);
    frame t of utterance b copies phoneme `#{i : cumsum_i <= t}`; frames past the utterance total are
    zero; `mel_len` is the UNCROPPED total; a `max_len` smaller than a total crops (negative F.pad).
    Returns (idx int64 (B,Tout) with -1 for zero rows, mel_len int64 (B,))."""
    di = duration.to(torch.float64).clamp(min=0).trunc().to(torch.int64)
    cs = di.cumsum(1)
    mel_len = cs[:, -1].clone()
    T = int(max_len) if max_len else int(mel_len.max())
    t = torch.arange(T, device=duration.device)[None, :].expand(duration.shape[0], T).contiguous()
    idx = torch.searchsorted(cs, t, right=True)
    idx = torch.where(t < mel_len[:, None], idx, torch.full_like(idx, -1))
    return idx, mel_len


def length_regulator(x, duration, max_len=None):
    idx, mel_len = length_regulator_index(duration, max_len)
    g = torch.gather(x, 1, idx.clamp(min=0)[..., None].expand(-1, -1, x.shape[2]))
    return g * (idx >= 0)[..., None].to(x.dtype), mel_len


def variance_adaptor(sd, x, spk, src_pad, max_len, pitch_t, energy_t, dur_t, controls, mc, train):
    """reference: fs_two/model/modules.py:142-217 (order of operations matters, see Appendix B):
    duration predictor sees x BEFORE the speaker embedding; energy predictor sees the pitch embedding."""
    p_c, e_c, d_c = controls
    pv = mc["variance_predictor"]["dropout"]
    va = "variance_adaptor."
    logd = variance_predictor(sd, va + "duration_predictor.", x, src_pad, pv, train)
    x = x + spk
    pitch = variance_predictor(sd, va + "pitch_predictor.", x, src_pad, pv, train)
    if pitch_t is not None:                                   # modules.py:92-101
        pidx = torch.bucketize(pitch_t, sd[va + "pitch_bins"])
    else:
        pitch = pitch * p_c
        pidx = torch.bucketize(pitch, sd[va + "pitch_bins"])
    x = x + F.embedding(pidx, sd[va + "pitch_embedding.weight"])
    energy = variance_predictor(sd, va + "energy_predictor.", x, src_pad, pv, train)
    if energy_t is not None:                                  # modules.py:131-140
        eidx = torch.bucketize(energy_t, sd[va + "energy_bins"])
    else:
        energy = energy * e_c
        eidx = torch.bucketize(energy, sd[va + "energy_bins"])
    x = x + F.embedding(eidx, sd[va + "energy_embedding.weight"])
    if dur_t is not None:
        d_rounded = dur_t
        x, mel_len = length_regulator(x, dur_t, max_len)
        mel_pad = None
    else:                                                     # modules.py:199-205
        d_rounded = torch.clamp(torch.round(torch.exp(logd) - 1) * d_c, min=0)
        x, mel_len = length_regulator(x, d_rounded, max_len)
        mel_pad = mask_from_lengths(mel_len)
    return x, pitch, energy, logd, d_rounded, mel_len, mel_pad


# ----------------------------------------------------------------------------- postnet / model

def postnet(sd, mel, train, bn_buffers=None):
    """reference: fs_two/transformer/Layers.py:71-143 — 5× Conv1d k5 + BatchNorm1d (+tanh except last),
    dropout 0.5 after every layer (F.dropout(..., self.training)).  Train mode uses batch statistics over
    all B·T positions, PAD rows included.  `bn_buffers` (dict) receives the updated running stats."""
    x = mel.transpose(1, 2)
    n = 0
    while ("postnet.convolutions.%d.0.conv.weight" % n) in sd:
        n += 1
    for i in range(n):
        pre = "postnet.convolutions.%d." % i
        w = sd[pre + "0.conv.weight"]
        x = F.conv1d(x, w, sd[pre + "0.conv.bias"], padding=(w.shape[2] - 1) // 2)
        rm, rv = sd[pre + "1.running_mean"], sd[pre + "1.running_var"]
        if train and bn_buffers is not None:
            rm, rv = rm.clone(), rv.clone()
        x = F.batch_norm(x, None if (train and bn_buffers is None) else rm,
                         None if (train and bn_buffers is None) else rv,
                         sd[pre + "1.weight"], sd[pre + "1.bias"], training=train, momentum=0.1, eps=1e-5)
        if train and bn_buffers is not None:
            bn_buffers[pre + "1.running_mean"], bn_buffers[pre + "1.running_var"] = rm, rv
        if i < n - 1:
            x = torch.tanh(x)
        x = _drop(x, 0.5, train)
    return x.transpose(1, 2)


def fs2_forward(sd, mc, speakers, texts, src_lens, max_src_len, mels=None, mel_lens=None, max_mel_len=None,
                e_targets=None, d_targets=None, pitches_raw=None, pitches_cwt=None, pitches_mean=None,
                pitches_std=None, p_control=1.0, e_control=1.0, d_control=1.0, train=False, bn_buffers=None):
    """reference: fs_two/model/fastspeech2.py:43-119 — returns the same 12-tuple."""
    src_pad = mask_from_lengths(src_lens, max_src_len)
    mel_pad = mask_from_lengths(mel_lens, max_mel_len) if mel_lens is not None else None
    x = encoder(sd, texts, src_pad, mc, train)
    spk = F.embedding(speakers, sd["speaker_emb.weight"])[:, None, :]
    x, pitch, energy, logd, d_rounded, mel_lens_out, mel_pad2 = variance_adaptor(
        sd, x, spk, src_pad, max_mel_len, pitches_raw, e_targets, d_targets,
        (p_control, e_control, d_control), mc, train)
    if mel_pad is None:
        mel_pad = mel_pad2
    x, mel_pad = decoder(sd, x, mel_pad, mc, train)
    mel = F.linear(x, sd["mel_linear.weight"], sd["mel_linear.bias"])
    post = postnet(sd, mel, train, bn_buffers) + mel
    return (mel, pitch, energy, logd, d_rounded, src_pad, mel_pad, src_lens, mel_lens_out, post, None, None)


def fs2_loss(batch, out):
    """reference: fs_two/model/loss.py:24-134 (use_cwt False).  Mel terms average over ALL B·T·80
    elements after zeroing PAD rows of prediction and target; total has shape (1,)."""
    mel_t, _, _, energy_t, dur_t, pitch_t = batch[6:12]
    mel, pitch, energy, logd, _, src_pad, mel_pad, _, _, post, _, _ = out
    src_ok, mel_ok = ~src_pad, ~mel_pad
    logd_t = torch.log(dur_t.float() + 1)
    mel_t = mel_t[:, : mel_ok.shape[1], :] * mel_ok[..., None]
    mel = mel * mel_ok[..., None]
    post = post * mel_ok[..., None]
    mel_total = F.mse_loss(mel, mel_t) + F.l1_loss(mel, mel_t) + F.l1_loss(post, mel_t)
    pitch_l = F.mse_loss(pitch.masked_select(src_ok), pitch_t.masked_select(src_ok))
    energy_l = F.mse_loss(energy.masked_select(src_ok), energy_t.masked_select(src_ok))
    dur_l = F.mse_loss(logd.masked_select(src_ok), logd_t.masked_select(src_ok))
    zero = torch.tensor([0])
    total = mel_total + dur_l + pitch_l + energy_l + zero + zero
    return total, mel_total, pitch_l, energy_l, dur_l, zero, zero


# ----------------------------------------------------------------------------- optimiser / train step

def lr_at(step, d_model=256, warmup=4000, anneal_steps=(300000, 400000, 500000), anneal_rate=0.7):
    """reference: fs_two/model/optimizer.py:35-53 — `step` is the value AFTER the increment."""
    lr = min(step ** -0.5, warmup ** -1.5 * step)
    for s in anneal_steps:
        if step > s:
            lr *= anneal_rate
    return d_model ** -0.5 * lr


def trainable_keys(sd):
    """Keys of `model.parameters()` that receive a gradient with use_cwt False: everything except the
    fixed tables, BN buffers and the unused CWT heads."""
    skip = ("position_enc", "pitch_bins", "energy_bins", "running_mean", "running_var",
            "num_batches_tracked", "variance_adaptor.pitch_mean.", "variance_adaptor.pitch_std.")
    return [k for k in sd if not any(s in k for s in skip)]


class OracleTrainer:
    """reference: train.py:24-56 + fs_two/model/optimizer.py:5-53 + torch.optim.Adam semantics
    (betas (0.95, 0.999), eps 1e-5, no weight decay, bias correction, grads None are skipped)."""

    def __init__(self, sd, mc, tc, current_step=0):
        self.sd = {k: v.clone() for k, v in sd.items()}
        self.mc, self.tc = mc, tc
        self.keys = trainable_keys(self.sd)
        for k in self.keys:
            self.sd[k].requires_grad_(True)
        self.m = {k: torch.zeros_like(self.sd[k]) for k in self.keys}
        self.v = {k: torch.zeros_like(self.sd[k]) for k in self.keys}
        self.t = 0
        self.current_step = current_step

    def train_step(self, batch, step, train_mode=True):
        opt = self.tc["optimizer"]
        acc = opt["grad_acc_step"]
        buffers = {}
        out = fs2_forward(self.sd, self.mc, *batch[2:], train=train_mode, bn_buffers=buffers)
        losses = fs2_loss(batch, out)
        (losses[0] / acc).sum().backward()
        with torch.no_grad():
            for k, v in buffers.items():
                self.sd[k].copy_(v)
            for k in self.sd:
                if k.endswith("num_batches_tracked"):
                    self.sd[k] += 1
        vals = [float(l.sum()) / acc for l in losses[1:]]
        if step % acc == 0:
            self.optimizer_step()
        return vals, out

    def grad_norm(self):
        return math.sqrt(sum(float(self.sd[k].grad.double().pow(2).sum()) for k in self.keys
                             if self.sd[k].grad is not None))

    def optimizer_step(self):
        opt = self.tc["optimizer"]
        b1, b2 = opt["betas"]
        eps, clip = opt["eps"], opt["grad_clip_thresh"]
        with torch.no_grad():
            coef = min(1.0, clip / (self.grad_norm() + 1e-6))       # nn.utils.clip_grad_norm_
            self.current_step += 1
            lr = lr_at(self.current_step, self.mc["transformer"]["encoder_hidden"], opt["warm_up_step"],
                       opt["anneal_steps"], opt["anneal_rate"])
            self.t += 1
            bc1, bc2 = 1 - b1 ** self.t, 1 - b2 ** self.t
            for k in self.keys:
                p = self.sd[k]
                if p.grad is None:
                    continue
                g = p.grad * coef
                self.m[k].mul_(b1).add_(g, alpha=1 - b1)
                self.v[k].mul_(b2).addcmul_(g, g, value=1 - b2)
                denom = (self.v[k].sqrt() / math.sqrt(bc2)).add_(eps)
                p.addcdiv_(self.m[k], denom, value=-lr / bc1)
                p.grad = None
