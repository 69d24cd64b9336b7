"""The train step with the reference's signature, running the sync-free native path.

reference: train.py:24-56 (`main_train_step`), fs_two/utils/tools.py:15-83 (`to_device`), fs_two/utils/model.py:12-38
(`get_model`), train.py:212-227 (checkpoint layout).
"""
import os

import numpy as np
import torch

from . import ops
from .fastspeech2 import FastSpeech2
from .loss import FastSpeech2Loss
from .optimizer import ScheduledOptim


def _t(x):
    return torch.from_numpy(x) if isinstance(x, np.ndarray) else torch.as_tensor(x)


def to_device(data, device="cuda:0", non_blocking=False):
    """reference: fs_two/utils/tools.py:15-83 — numpy/CPU batch tuple -> device tensors with the reference dtypes
    (speakers/texts/durations long, mels/pitches float, lens as stored; NaNs in pitches_cwt -> 0)."""
    nb = non_blocking
    if len(data) == 15:
        (ids, raw_texts, speakers, texts, src_lens, max_src_len, mels, mel_lens, max_mel_len, energies, durations,
         pitches_raw, pitches_cwt, pitches_mean, pitches_std) = data
        mv = lambda t: t.to(device, non_blocking=nb)
        return (ids, raw_texts, mv(_t(speakers).long()), mv(_t(texts).long()), mv(_t(src_lens)), max_src_len, mv(_t(mels).float()),
                mv(_t(mel_lens)), max_mel_len, mv(_t(energies)), mv(_t(durations).long()), mv(_t(pitches_raw).float()),
                mv(torch.nan_to_num(_t(pitches_cwt).float(), nan=0.0)), mv(_t(pitches_mean).float()), mv(_t(pitches_std).float()))
    if len(data) == 6:
        ids, raw_texts, speakers, texts, src_lens, max_src_len = data
        return (ids, raw_texts, _t(speakers).long().to(device), _t(texts).long().to(device), _t(src_lens).to(device), max_src_len)
    raise ValueError("batch tuple must have 15 (train) or 6 (inference) entries, got %d" % len(data))


def get_model(cfg, device, train=False):
    """reference: fs_two/utils/model.py:12-38.  `cfg.tts.load_path` (absent from the shipped config) is optional;
    the speaker embedding is re-inserted with the rule of fsapi.py:28-30 (the reference's training-resume path
    silently drops it)."""
    model = FastSpeech2(cfg.preprocess_config, cfg.model_config, device=device,
                        seed=int(cfg.get("mi355x", {}).get("seed", 1234)) if hasattr(cfg, "get") else 1234)
    load_path = cfg.tts.get("load_path") if hasattr(cfg.tts, "get") else None
    ckpt = None
    if load_path:
        ckpt = torch.load(load_path, map_location="cpu")
        state = dict(ckpt["model"])
        if "embedding" in ckpt:
            state["speaker_emb.weight"] = ckpt["embedding"]
        model.load_state_dict(state, strict=False)
    if train:
        model.train()
        optim = ScheduledOptim(model, cfg.train_config, cfg.model_config, cfg.tts.restore_step)
        if ckpt is not None and isinstance(ckpt.get("optimizer"), dict):
            # the reference saves torch.optim.Adam's state_dict (train.py:221) but never restores it (row f-4); both that
            # layout and this build's earlier flat one are accepted
            optim.load_state_dict(ckpt["optimizer"])
        return model, optim
    model.eval()
    return model


def save_checkpoint(model, optimizer, path):
    """reference: train.py:212-227 — {"model": state_dict minus speaker_emb.weight, "embedding": ..., "optimizer": ...}."""
    sd = {k: v.detach().cpu().contiguous() for k, v in model.state_dict().items()}
    emb = sd.pop("speaker_emb.weight")
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    torch.save({"model": sd, "embedding": emb, "optimizer": optimizer.state_dict() if optimizer is not None else None}, path)


def main_train_step(model, batch, step, optimizer, cfg, Loss, reducer=None):
    """reference: train.py:24-56.  forward -> loss -> backward (grads / grad_acc_step) -> every grad_acc_step-th call:
    clip, LR update, Adam, zero_grad.  Returns (losses[6 floats], output 12-tuple) like the reference.

    Everything up to the final read of the loss values is enqueued without host synchronisation; `reducer`
    (tts_king_amd.parallel.GradReducer) all-reduces gradient buckets on a side stream while backward runs."""
    grad_acc_step = cfg.train_config["optimizer"]["grad_acc_step"]
    if not model.training:
        model.train()
    dev = model.device
    with torch.no_grad():
        out, ctx = model._forward(True, batch[2].to(dev).long().contiguous(), batch[3].to(dev).long().contiguous(),
                                  batch[4].to(dev).long().contiguous(), int(batch[5]), batch[7], batch[8], batch[9], batch[10],
                                  batch[11], 1.0, 1.0, 1.0)
        mel, pitch, energy, logd, d_rounded, src_masks, mel_masks, mel_lens_out, post = out
        targets = Loss.targets_of(batch, dev)
        losses, dmel_sum, dpost, dp, de, dd = ops.fs2_loss(mel, post, targets[0], targets[1], pitch, energy, logd, targets[2],
                                                           targets[3], targets[4], targets[5], grad_scale=1.0 / grad_acc_step)
        do_step = step % grad_acc_step == 0
        if reducer is not None and do_step:
            model.backward_native(ctx, dmel_sum, dpost, dp, de, dd, on_bucket=reducer.on_group_done)
            reducer.finish()
        else:
            model.backward_native(ctx, dmel_sum, dpost, dp, de, dd)
        if do_step:
            optimizer.step_and_update_lr(advance_rng=True)          # the end-of-step dropout-counter tick rides along
            optimizer.zero_grad()
        else:
            ops.rng_advance(model._state())
    output = (mel, pitch, energy, logd, d_rounded, src_masks, mel_masks, batch[4], mel_lens_out, post, None, None)
    vals = losses.cpu().tolist()                       # the step's only host read (reference: 6x .item(), train.py:45)
    return [v / grad_acc_step for v in vals[1:7]], output
