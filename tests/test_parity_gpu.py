"""GPU: the parity cases VERDICT r01 asked for on top of tests/test_fs2_gpu.py — full-size configs[1] against the oracle,
the reference's default `grad_acc_step: 4` cycle, a 20-step trajectory, the train-mode decoder truncation at
max_seq_len, dropout keep-fraction / scale at every kind of dropout site, and the drop-in call sites the advisor named
(`loss.backward()` advances the dropout counters, `.to()` + `load_state_dict` refreshes the bf16 shadow, synthesizer
outputs are not aliased).

Stated tolerances (bf16 storage / fp32 accumulate vs the oracle's fp32), exactly as asserted below: losses rel 1 %; global
gradient norm rel 2 %; per-parameter-group gradient norm rel 6 %; whole gradient tensors rel-RMS 8 %; 20-step trajectory: the
total loss within 3 % of the oracle's at all but two steps and within 8 % at those, 1.5 % on average and over the last five
steps, every component within 30 % (the reasons are at the assertions)."""
import copy
import math

import numpy as np
import pytest
import torch

from oracle import fs2 as ofs2
from tests.oracle_util import fs2_state_dict, rel_rms
from tts_king_amd.synthetic import make_batch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def build(cfg, weight_seed, dropout=True, n_speakers=65):
    from tts_king_amd.fastspeech2 import FastSpeech2
    m = FastSpeech2(cfg.preprocess_config, cfg.model_config, n_speakers, device=DEV)
    m.load_state_dict(fs2_state_dict(cfg, weight_seed))
    if not dropout:
        m.p_enc = m.p_dec = m.p_var = m.p_post = 0.0
    return m


def no_dropout_config(cfg):
    mc0 = copy.deepcopy(cfg.model_config)
    mc0["transformer"]["encoder_dropout"] = mc0["transformer"]["decoder_dropout"] = 0.0
    mc0["variance_predictor"]["dropout"] = 0.0
    return mc0


class oracle_without_dropout:
    """The oracle's PostNet dropout is hard-coded (0.5, as in the reference: Layers.py:137-141): switch every site off."""

    def __enter__(self):
        self.keep = ofs2._drop
        ofs2._drop = lambda x, p, train: x

    def __exit__(self, *exc):
        ofs2._drop = self.keep
        return False


GROUPS = ("encoder.src_word_emb", "encoder.layer_stack.0", "encoder.layer_stack.1", "encoder.layer_stack.2", "encoder.layer_stack.3",
          "speaker_emb", "variance_adaptor.duration_predictor", "variance_adaptor.pitch_predictor", "variance_adaptor.energy_predictor",
          "variance_adaptor.pitch_embedding", "variance_adaptor.energy_embedding", "decoder.layer_stack.0", "decoder.layer_stack.1",
          "decoder.layer_stack.2", "decoder.layer_stack.3", "decoder.layer_stack.4", "decoder.layer_stack.5", "mel_linear",
          "postnet.convolutions.0", "postnet.convolutions.1", "postnet.convolutions.2", "postnet.convolutions.3", "postnet.convolutions.4")


class oracle_with_masks:
    """Run the oracle with the dropout keep-masks of the HIP path: `masks` = the (keep uint8 tensor shaped like the oracle's operand,
    p) pairs in the order the oracle reaches its dropout sites (`ofs2._drop` calls)."""

    def __init__(self, masks):
        self.masks, self.pos = masks, 0

    def __enter__(self):
        self.keep = ofs2._drop

        def drop(x, p, train):
            if not (train and p > 0):
                return x
            k, pk = self.masks[self.pos]
            self.pos += 1
            assert tuple(k.shape) == tuple(x.shape) and abs(pk - p) < 1e-9, (self.pos, k.shape, x.shape, pk, p)
            return x * k.to(x.dtype) * (1.0 / (1.0 - p))
        ofs2._drop = drop
        return self

    def __exit__(self, *exc):
        ofs2._drop = self.keep
        return False


def hip_dropout_masks(m, B, L, T):
    """The keep-masks the HIP step draws at the model's CURRENT dropout state (seed, step), for every site of one training forward,
    in the oracle's call order and operand layouts: encoder blocks (fc, w_2) -> duration / pitch / energy predictors (two each) ->
    decoder blocks -> PostNet layers (the oracle's PostNet runs channels-first: (B, C, T))."""
    from tts_king_amd import ops
    st = m._state()
    d, nm = m.d, m.n_mel
    out = []

    def site(s, rows_shape, C, p):
        k = ops.dropout_keep_mask(st, s, rows_shape[0] * rows_shape[1] * C, p).view(rows_shape[0], rows_shape[1], C).cpu()
        return k, p
    for i in range(m.n_enc):
        out += [site(2 * i, (B, L), d, m.p_enc), site(2 * i + 1, (B, L), d, m.p_enc)]
    Fh = m.model_config["variance_predictor"]["filter_size"]
    for g in range(3):
        out += [site(200 + 2 * g, (B, L), Fh, m.p_var), site(201 + 2 * g, (B, L), Fh, m.p_var)]
    for i in range(m.n_dec):
        out += [site(100 + 2 * i, (B, T), d, m.p_dec), site(101 + 2 * i, (B, T), d, m.p_dec)]
    for i in range(5):
        C = nm if i == 4 else 512
        k, p = site(300 + i, (B, T), C, m.p_post)
        out.append((k.transpose(1, 2).contiguous(), p))
    return out


@pytest.mark.parametrize("dropout", [False, True])
def test_full_size_step_vs_oracle(cfg, dropout):
    """BASELINE.json configs[1] at full size (B=16, L=64, T=423, 65 speakers): forward, loss and backward against the oracle on the same
    batch — the four losses within 1 %, the global gradient norm within 2 %, the gradient norm of every parameter group (FFT block,
    predictor, embedding table, PostNet layer) within 6 %, ten whole gradient tensors within 8 % rel-RMS.
    dropout = False: every site off on both sides.  dropout = True (what bench.py times: 31 sites, reference SubLayers.py:62,98,
    modules.py:283,295, Layers.py:137-140): the oracle runs with the very keep-masks the HIP kernels draw (ttsk_dropout_keep_mask at
    the step's (seed, step): masks are functions of (seed, step, site, element), regenerated in backward, never stored) — so this
    also pins every kernel's site number and element indexing, forward and backward."""
    from tts_king_amd import ops
    m = build(cfg, 7, dropout=dropout).train()
    b = make_batch(16, 64, seed=1234)
    assert int(b[8]) == 423 and int(b[7].sum()) == 6070
    dev_b = [t.to(DEV) if torch.is_tensor(t) else t for t in b]
    masks = hip_dropout_masks(m, 16, 64, 423) if dropout else None
    with torch.no_grad():
        out, ctx = m._forward(True, dev_b[2], dev_b[3], dev_b[4], int(b[5]), dev_b[7], b[8], dev_b[9], dev_b[10], dev_b[11], 1.0, 1.0, 1.0)
        losses, dmel_sum, dpost, dp, de, dd = ops.fs2_loss(out[0], out[8], dev_b[6], dev_b[7], out[1], out[2], out[3], dev_b[11],
                                                           dev_b[9], dev_b[10], dev_b[4], grad_scale=1.0)
        m.backward_native(ctx, dmel_sum, dpost, dp, de, dd)
    torch.cuda.synchronize()
    got = losses.cpu().tolist()
    sd = fs2_state_dict(cfg, 7)
    tr = ofs2.OracleTrainer(sd, copy.deepcopy(cfg.model_config) if dropout else no_dropout_config(cfg), cfg.train_config, 0)
    with (oracle_with_masks(masks) if dropout else oracle_without_dropout()) as feeder:
        o = ofs2.fs2_forward(tr.sd, tr.mc, *b[2:], train=True, bn_buffers={})
        ls = ofs2.fs2_loss(b, o)
        ls[0].sum().backward()
    if dropout:
        assert feeder.pos == len(masks) == 31, (feeder.pos, len(masks))
    want = [float(l.sum()) for l in ls]
    print("losses HIP", [round(v, 5) for v in got[:5]], "oracle", [round(v, 5) for v in want[:5]])
    np.testing.assert_allclose(got[1:5], want[1:5], rtol=0.01)
    np.testing.assert_allclose(got[0], want[0], rtol=0.01)
    r = rel_rms(out[0].float().cpu(), o[0].detach())
    print("full-size train mel rel-RMS %.3f%%" % (100 * r))
    assert r <= 0.01
    named = dict(m.named_parameters())
    gsq, osq = 0.0, 0.0
    worst = (0.0, None)
    for grp in GROUPS:
        a = math.sqrt(sum(float(named[k].grad.double().pow(2).sum()) for k in tr.keys if k.startswith(grp + ".")))
        w = math.sqrt(sum(float(tr.sd[k].grad.double().pow(2).sum()) for k in tr.keys if k.startswith(grp + ".")))
        gsq, osq = gsq + a * a, osq + w * w
        err = abs(a - w) / w
        print("  group %-42s |g| HIP %.5f oracle %.5f  (%.2f%%)" % (grp, a, w, 100 * err))
        if err > worst[0]:
            worst = (err, grp)
    # whole gradient tensors at both ends of the backward chain (direction, not only size)
    rels = {}
    for k in ("postnet.convolutions.4.0.conv.weight", "mel_linear.weight", "decoder.layer_stack.5.pos_ffn.w_1.weight",
              "decoder.layer_stack.0.slf_attn.w_qs.weight", "variance_adaptor.pitch_predictor.conv_layer.conv1d_1.conv.weight",
              "variance_adaptor.energy_embedding.weight", "speaker_emb.weight", "encoder.layer_stack.3.pos_ffn.w_2.weight",
              "encoder.layer_stack.0.slf_attn.fc.weight", "encoder.src_word_emb.weight"):
        r = rel_rms(named[k].grad.float().cpu(), tr.sd[k].grad)
        print("  grad %-64s rel-RMS vs oracle %.2f%%" % (k, 100 * r))
        rels[k] = r
    bars = {k: 0.08 for k in rels}
    if dropout:
        # With half of the predictors' and the PostNet's activations dropped (p = 0.5), a whole-tensor comparison is at the mercy of the
        # reference's own soft spots: e.g. train-mode BatchNorm divides by batch deviations that shrink with the kept half, and a 2^-9
        # rounding of the operands turns into > 10 % of the last PostNet conv's weight gradient (whose NORM stays within 0.1 %).
        # Calibrate instead of guessing: the oracle against ITSELF with its matrices rounded to bf16 — the HIP path's operand precision
        # — under the same masks; per tensor the bar is max(8 %, 1.5 x that).
        sd16 = {k: (v.to(torch.bfloat16).float() if (v.is_floating_point() and v.dim() >= 2) else v.clone()) for k, v in sd.items()}
        tr16 = ofs2.OracleTrainer(sd16, copy.deepcopy(cfg.model_config), cfg.train_config, 0)
        with oracle_with_masks(masks):
            o16 = ofs2.fs2_forward(tr16.sd, tr16.mc, *b[2:], train=True, bn_buffers={})
            ofs2.fs2_loss(b, o16)[0].sum().backward()
        for k in rels:
            own = rel_rms(tr16.sd[k].grad, tr.sd[k].grad)
            bars[k] = max(0.08, 1.5 * own)
            print("  grad %-64s oracle with bf16-rounded matrices vs oracle: rel-RMS %.2f%% -> bar %.1f%%" % (k, 100 * own, 100 * bars[k]))
    for k, r in rels.items():
        assert r <= bars[k], (k, r, bars[k])
    gn, on = math.sqrt(gsq), math.sqrt(osq)
    assert abs(on - tr.grad_norm()) <= 1e-6 * on          # the groups cover every trainable key
    print("global grad norm HIP %.5f oracle %.5f; worst group %s" % (gn, on, worst))
    assert abs(gn - on) <= 0.02 * on
    assert worst[0] <= 0.06, worst


@pytest.mark.parametrize("dropout", [False, True])
def test_grad_acc_step_4_cycle_vs_oracle(cfg, dropout):
    """The reference's default `grad_acc_step: 4` (config.yaml:51, train.py:33,43-54): four micro-steps on four different
    batches accumulate (loss / 4) gradients, the fourth call clips, updates the LR, runs Adam and zeroes the gradients —
    against OracleTrainer on the same four batches.  dropout = True (the reference's own configuration): the oracle takes, micro-step
    by micro-step, the keep-masks the HIP path is about to draw — the dropout counter ticks after EVERY micro-step, update or not."""
    from tts_king_amd.loss import FastSpeech2Loss
    from tts_king_amd.optimizer import ScheduledOptim
    from tts_king_amd.train_step import main_train_step, to_device
    c = copy.deepcopy(cfg)
    assert c.train_config["optimizer"]["grad_acc_step"] == 4
    m = build(c, 7, dropout=dropout)
    opt = ScheduledOptim(m, c.train_config, c.model_config, 1000)
    loss_fn = FastSpeech2Loss(c.preprocess_config, c.model_config)
    sd0 = fs2_state_dict(c, 7)
    tr = ofs2.OracleTrainer(sd0, copy.deepcopy(c.model_config) if dropout else no_dropout_config(c), c.train_config, current_step=1000)
    batches = [make_batch(3, 40 + 8 * i, seed=50 + i, ragged=True) for i in range(4)]
    flat0 = m.flat_buffers()[0].clone()
    seen = []
    for step in range(1, 5):
        b = batches[step - 1]
        masks = hip_dropout_masks(m, 3, int(b[5]), int(b[8])) if dropout else None
        if dropout:
            key = masks[-1][0][:, :, :32].clone()               # (the PostNet's last site: the same shape prefix in every batch)
            assert all(not torch.equal(key, k) for k in seen), "micro-step %d drew an earlier micro-step's masks" % step
            seen.append(key)
        vals, _ = main_train_step(m, to_device(b, DEV), step, opt, c, loss_fn)
        with (oracle_with_masks(masks) if dropout else oracle_without_dropout()) as feeder:
            ovals, _ = tr.train_step(b, step)
        if dropout:
            assert feeder.pos == 31
        print("micro-step %d losses" % step, [round(v, 5) for v in vals[:4]], [round(v, 5) for v in ovals[:4]])
        np.testing.assert_allclose(vals[:4], ovals[:4], rtol=0.01)          # both report loss / grad_acc_step
        if step < 4:
            assert torch.equal(m.flat_buffers()[0], flat0), "weights moved before the 4th micro-step"
            assert opt.current_step == 1000
            assert float(m.flat_buffers()[1].abs().max()) > 0.0            # gradients are accumulating
    assert opt.current_step == 1001 and tr.current_step == 1001
    assert abs(opt.lr() - ofs2.lr_at(1001)) < 1e-12
    assert float(m.flat_buffers()[1].abs().max()) == 0.0
    # accumulated-gradient norm (what the clip saw) and the update itself
    cos_min, worst = 1.0, None
    for k in tr.keys:
        if "w_ks.bias" in k or ("postnet" in k and k.endswith("conv.bias")):
            continue                                                         # true gradient 0: Adam normalises pure noise
        mine = (m.get(k).detach().cpu() - sd0[k]).flatten().double()
        ref = (tr.sd[k].detach() - sd0[k]).flatten().double()
        cos = float((mine @ ref) / (mine.norm() * ref.norm() + 1e-30))
        if cos < cos_min:
            cos_min, worst = cos, k
        assert abs(float(mine.norm()) / float(ref.norm()) - 1) < 0.1, k
    print("grad_acc 4: min cosine(update, oracle update) %.4f at %s" % (cos_min, worst))
    assert cos_min > 0.9


def test_trajectory_20_steps_vs_oracle(cfg):
    """20 consecutive full train steps (fwd, loss, bwd, clip, Adam with the warm-up LR) on 4 alternating B=2 batches, dropout
    off, HIP vs oracle from the same initial weights, at scheduler step 100 (lr 2.5e-5: the oracle's total loss falls from
    12.2 to 4.7 over the 20 steps — real learning, not the chaotic blow-up a near-peak LR gives random weights): the total
    loss stays within 3 % of the oracle's at all but two steps and within 8 % at those (1.5 % on average, 1.5 % over the last five steps) and every component within 30 % (the smallest one, the duration loss, wanders most), with no growth along
    the curve: the per-step bf16 differences do not compound."""
    from tts_king_amd.loss import FastSpeech2Loss
    from tts_king_amd.optimizer import ScheduledOptim
    from tts_king_amd.train_step import main_train_step, to_device
    c = copy.deepcopy(cfg)
    c.train_config["optimizer"]["grad_acc_step"] = 1
    m = build(c, 7, dropout=False)
    opt = ScheduledOptim(m, c.train_config, c.model_config, 100)
    loss_fn = FastSpeech2Loss(c.preprocess_config, c.model_config)
    tr = ofs2.OracleTrainer(fs2_state_dict(c, 7), no_dropout_config(c), c.train_config, current_step=100)
    batches = [make_batch(2, 48, seed=300 + i, ragged=True) for i in range(4)]
    dev_batches = [to_device(b, DEV) for b in batches]
    curve = []
    for s in range(20):
        vals, _ = main_train_step(m, dev_batches[s % 4], s + 1, opt, c, loss_fn)
        with oracle_without_dropout():
            ovals, _ = tr.train_step(batches[s % 4], s + 1)
        tot, otot = sum(vals[:4]), sum(ovals[:4])
        err = max(abs(a - w) / abs(w) for a, w in zip(vals[:4], ovals[:4]))
        curve.append((tot, otot, err))
        print("step %2d total HIP %.5f oracle %.5f (%.2f%%)  worst component %.2f%%  HIP %s oracle %s"
              % (s + 1, tot, otot, 100 * abs(tot - otot) / otot, 100 * err, [round(v, 4) for v in vals[:4]], [round(v, 4) for v in ovals[:4]]))
    tot_err = [abs(a - b) / b for a, b, _ in curve]
    # Adam's first updates are sign-like (m / sqrt(v) = +-1 per weight), so bf16-level gradient noise on near-zero gradient
    # elements shows up in the fastest-falling component (duration: 2.15 -> 0.28 in 20 steps) for a few steps and then decays:
    # the bar is on the total at every step, on every component loosely, and on the end of the curve tightly
    # measured over the round's builds (every change of a kernel's summation order re-rolls the noise): worst step of the total
    # 2.3-5.1 % (always steps 14-15, where both curves spike on the same batch; 16+ are back under 1.5 %), mean 0.7-1.1 %, last five 0.6-0.9 %, worst component 10-22 % — always the duration loss, the smallest term
    # (0.3 of a total of 4.8 by then, falling 8-fold over the run): 22 % of it is 1.4 % of the total
    assert sorted(tot_err)[-3] <= 0.03 and max(tot_err) <= 0.08, tot_err     # all but two steps within 3 %, the spike (steps 14-15) within 8 %
    assert max(e for _, _, e in curve) <= 0.30, curve
    assert sum(tot_err) / len(tot_err) <= 0.015 and sum(tot_err[-5:]) / 5 <= 0.015, tot_err
    assert curve[-1][1] < 0.5 * curve[0][1], "the oracle's loss did not go down: the trajectory test is not exercising learning"
    assert opt.current_step == 120


def test_trajectory_with_dropout_on_vs_oracle(cfg):
    """Eight consecutive full train steps WITH dropout (what the trainer and bench.py run), the oracle fed, step by step, the keep-masks
    the HIP kernels are about to draw (ttsk_dropout_keep_mask at the device's current (seed, step)): the masks change from step to step
    (the end-of-step tick of the dropout counter rides in the optimizer launch), all 31 sites are consumed in every step, and the losses
    follow the oracle's as closely as in the dropout-off trajectory."""
    from tts_king_amd.loss import FastSpeech2Loss
    from tts_king_amd.optimizer import ScheduledOptim
    from tts_king_amd.train_step import main_train_step, to_device
    c = copy.deepcopy(cfg)
    c.train_config["optimizer"]["grad_acc_step"] = 1
    m = build(c, 7, dropout=True)
    opt = ScheduledOptim(m, c.train_config, c.model_config, 100)
    loss_fn = FastSpeech2Loss(c.preprocess_config, c.model_config)
    tr = ofs2.OracleTrainer(fs2_state_dict(c, 7), copy.deepcopy(c.model_config), c.train_config, current_step=100)
    batches = [make_batch(2, 48, seed=300 + i, ragged=True) for i in range(2)]
    dev_batches = [to_device(b, DEV) for b in batches]
    errs, prev_first = [], None
    for s in range(8):
        b = batches[s % 2]
        masks = hip_dropout_masks(m, 2, int(b[5]), int(b[8]))
        first = masks[0][0].clone()
        if prev_first is not None and prev_first.shape == first.shape:
            assert not torch.equal(prev_first, first), "step %d drew the masks of the step before: the dropout counter did not advance" % (s + 1)
        prev_first = first if s % 2 == 1 else prev_first      # (same batch shape two steps apart)
        vals, _ = main_train_step(m, dev_batches[s % 2], s + 1, opt, c, loss_fn)
        with oracle_with_masks(masks) as feeder:
            ovals, _ = tr.train_step(b, s + 1)
        assert feeder.pos == len(masks) == 31, (s, feeder.pos)
        tot, otot = sum(vals[:4]), sum(ovals[:4])
        comp = max(abs(a - w) / abs(w) for a, w in zip(vals[:4], ovals[:4]))
        errs.append(abs(tot - otot) / otot)
        print("step %d total HIP %.5f oracle %.5f (%.2f%%)  worst component %.2f%%" % (s + 1, tot, otot, 100 * errs[-1], 100 * comp))
        assert comp <= 0.30, (s, vals, ovals)
    assert max(errs) <= 0.05 and sum(errs) / len(errs) <= 0.02, errs
    assert opt.current_step == 108


def test_train_mode_truncates_decoder_at_max_seq_len(cfg):
    """reference: transformer/Models.py:172-180 (train mode: decoder input, mask and output cut to max_seq_len = 1000 frames)
    with loss.py:57-58 (mel targets cropped to the mask's length; `mel_lens` stay uncropped).  One utterance longer than
    1000 frames: shapes, masks and losses against the oracle."""
    from tts_king_amd import ops
    m = build(cfg, 7, dropout=False).train()
    b = make_batch(2, 200, seed=77, ragged=True)
    T_full = int(b[8])
    assert T_full > cfg.model_config["max_seq_len"], T_full
    dev_b = [t.to(DEV) if torch.is_tensor(t) else t for t in b]
    with torch.no_grad():
        out, ctx = m._forward(True, dev_b[2], dev_b[3], dev_b[4], int(b[5]), dev_b[7], b[8], dev_b[9], dev_b[10], dev_b[11], 1.0, 1.0, 1.0)
        losses, dmel_sum, dpost, dp, de, dd = ops.fs2_loss(out[0], out[8], dev_b[6], dev_b[7], out[1], out[2], out[3], dev_b[11],
                                                           dev_b[9], dev_b[10], dev_b[4], grad_scale=1.0)
        m.backward_native(ctx, dmel_sum, dpost, dp, de, dd)
    torch.cuda.synchronize()
    tr = ofs2.OracleTrainer(fs2_state_dict(cfg, 7), no_dropout_config(cfg), cfg.train_config, 0)
    with oracle_without_dropout():
        o = ofs2.fs2_forward(tr.sd, tr.mc, *b[2:], train=True, bn_buffers={})
        ls = ofs2.fs2_loss(b, o)
        ls[0].sum().backward()
    assert out[0].shape == (2, 1000, 80) and tuple(o[0].shape) == (2, 1000, 80)
    assert out[6].shape == (2, 1000) and torch.equal(out[6].cpu(), o[6])
    assert out[7].cpu().tolist() == o[8].tolist() == b[7].tolist()            # mel_lens uncropped
    got, want = losses.cpu().tolist(), [float(l.sum()) for l in ls]
    print("T=%d -> 1000: losses HIP" % T_full, [round(v, 5) for v in got[:5]], "oracle", [round(v, 5) for v in want[:5]])
    np.testing.assert_allclose(got[:2], want[:2], rtol=0.01)         # total and mel terms: means over 160,000 elements
    np.testing.assert_allclose(got[2:5], want[2:5], rtol=0.02)       # pitch / energy / duration: means over ~100 phonemes (bf16 noise 1 %)
    r = rel_rms(out[0].float().cpu(), o[0].detach())
    assert r <= 0.01, r
    named = dict(m.named_parameters())
    gn = math.sqrt(sum(float(named[k].grad.double().pow(2).sum()) for k in tr.keys))
    print("global grad norm HIP %.5f oracle %.5f" % (gn, tr.grad_norm()))
    assert abs(gn - tr.grad_norm()) <= 0.03 * tr.grad_norm()


# ------------------------------------------------------------------------------------------------ dropout sites

def _rnd(*shape, seed):
    return torch.randn(*shape, generator=torch.Generator().manual_seed(seed))


def test_batchnorm_dropout_site_keep_fraction_and_scale():
    """PostNet: F.dropout(tanh(BN(conv)), 0.5, training) (Layers.py:137-141).  With p = 0.5 every output element is either 0
    or exactly 2x the p = 0 output; the keep fraction is 1 - p within 4 sigma; the backward uses the same mask."""
    from tts_king_amd import ops
    rows, C, p = 4096, 512, 0.5
    x = _rnd(rows, C, seed=1).to(DEV)
    mean, rstd = x.mean(0), (x.var(0, unbiased=False) + 1e-5).rsqrt()
    gamma, beta = (1 + 0.1 * _rnd(C, seed=2)).to(DEV), (0.1 * _rnd(C, seed=3)).to(DEV)
    st = ops.optim_state(DEV, seed=5)
    rng = ops.rng_of(st)
    for use_tanh, site in ((True, 300), (False, 304)):
        base = ops.bn_apply(x, mean, rstd, gamma, beta, use_tanh, p=0.0, site=site, rng=rng).float()
        got = ops.bn_apply(x, mean, rstd, gamma, beta, use_tanh, p=p, site=site, rng=rng).float()
        keep = got != 0
        frac = float(keep.float().mean())
        sigma = math.sqrt(p * (1 - p) / keep.numel())
        assert abs(frac - (1 - p)) <= 4 * sigma + float((base == 0).float().mean()), (frac, sigma)
        torch.testing.assert_close(got[keep], (base / (1 - p))[keep], rtol=2 ** -7, atol=1e-6)     # one bf16 rounding apart
        again = ops.bn_apply(x, mean, rstd, gamma, beta, use_tanh, p=p, site=site, rng=rng).float()
        assert torch.equal(again, got)
        other = ops.bn_apply(x, mean, rstd, gamma, beta, use_tanh, p=p, site=site + 1, rng=rng).float()
        assert not torch.equal(other != 0, keep)
        # backward: gradient is zero exactly where the forward dropped
        dout = _rnd(rows, C, seed=4).to(DEV)
        dg, db = torch.zeros(C, device=DEV), torch.zeros(C, device=DEV)
        dx = ops.bn_bwd(dout, x, mean, rstd, gamma, beta, use_tanh, p=p, site=site, rng=rng, dgamma=dg, dbeta=db)
        # d(beta) = column sums of the masked, scaled upstream gradient (times tanh' when tanh is on)
        y = (x - mean) * rstd * gamma + beta
        dact = dout * keep.float() / (1 - p) * ((1 - torch.tanh(y) ** 2) if use_tanh else 1.0)
        torch.testing.assert_close(db, dact.sum(0), rtol=2e-2, atol=0.5)
        assert bool(torch.isfinite(dx.float()).all())


def test_predictor_dropout_site_keep_fraction_and_scale():
    """VariancePredictor: Dropout(0.5) AFTER LayerNorm (modules.py:286,298).  Output elements are 0 or 2x the p = 0 output,
    keep fraction 0.5 within 4 sigma, and the head (Linear(256,1)) sees the dropped activations."""
    from tts_king_amd import ops
    rows, D, p = 2048, 256, 0.5
    h = torch.relu(_rnd(rows, D, seed=6)).to(torch.bfloat16).to(DEV)
    gamma, beta = (1 + 0.1 * _rnd(D, seed=7)).to(DEV), (0.5 + 0.1 * _rnd(D, seed=8)).to(DEV)
    st = ops.optim_state(DEV, seed=11)
    rng = ops.rng_of(st)
    base, *_ = ops.layernorm_fwd(h, None, gamma, beta, None, 0, p_post=0.0, site_post=200, rng=rng, save_z=False)
    got, _, mean, rstd, _ = ops.layernorm_fwd(h, None, gamma, beta, None, 0, p_post=p, site_post=200, rng=rng, save_z=False)
    base, got = base.float(), got.float()
    keep = got != 0
    frac = float(keep.float().mean())
    sigma = math.sqrt(p * (1 - p) / keep.numel())
    assert abs(frac - (1 - p)) <= 4 * sigma + float((base == 0).float().mean()), (frac, sigma)
    torch.testing.assert_close(got[keep], (base / (1 - p))[keep], rtol=2 ** -7, atol=1e-6)
    # the fused head reads the dropped activations: head_out = sum_c w[c] * dropped[c] + b on valid rows
    w, bias = (_rnd(D, seed=9) * D ** -0.5).to(DEV), torch.tensor([0.1], device=DEV)
    lens = torch.full((rows // 64,), 64, dtype=torch.int64, device=DEV)
    _, _, _, _, ho = ops.layernorm_fwd(h, None, gamma, beta, lens, 64, p_post=p, site_post=200, rng=rng, save_z=False,
                                       head=(w, bias), want_out=False)
    y32 = torch.nn.functional.layer_norm(h.float(), (D,), gamma, beta)
    want = (y32 * keep.float() / (1 - p)) @ w + bias
    torch.testing.assert_close(ho, want, rtol=1e-3, atol=1e-2)
    # backward through the same mask: dz is zero wherever relu'(h) = 0; and non-trivial elsewhere
    dz, _, _, _ = ops.layernorm_bwd(_rnd(rows, D, seed=10).to(torch.bfloat16).to(DEV), h, mean, rstd, gamma, beta, None, 0,
                                    relu_in=True, p_post=p, site_post=200, rng=rng)
    assert float(dz.float()[h.float() == 0].abs().max()) == 0.0


# ------------------------------------------------------------------------------------------------ drop-in call sites

def test_bridge_path_advances_dropout_counters(cfg):
    """`model(*batch[2:])` -> `Loss(...)[0].backward()` (train.py:36-44) twice: the second step draws new dropout masks
    (Philox masks are functions of (seed, step, site, element); the counter ticks once per backward)."""
    from tts_king_amd.loss import FastSpeech2Loss
    m = build(cfg, 7, dropout=True).train()
    loss_fn = FastSpeech2Loss(cfg.preprocess_config, cfg.model_config)
    b = make_batch(2, 32, seed=3, ragged=True)
    mels = []
    for _ in range(2):
        step0 = int(m._state()[3])
        o = m(*b[2:])
        mels.append(o[9].detach().clone())
        loss_fn(b, o)[0].backward()
        torch.cuda.synchronize()
        assert int(m._state()[3]) == step0 + 1
    assert not torch.equal(mels[0], mels[1]), "two consecutive training forwards produced identical (same-mask) outputs"
    m.p_enc = m.p_dec = m.p_var = m.p_post = 0.0
    a = m(*b[2:])[9].detach().clone()
    assert torch.equal(a, m(*b[2:])[9].detach())          # and with dropout off the forward is a pure function of the weights


def test_shadow_refreshes_after_to_and_load_state_dict(cfg):
    """The reference idiom `FastSpeech2(...).to(device)` (fsapi.py:23, utils/model.py:27) followed by `load_state_dict`:
    the bf16 weight shadow the kernels read must follow the new weights."""
    from tts_king_amd.fastspeech2 import FastSpeech2
    m = FastSpeech2(cfg.preprocess_config, cfg.model_config, 65, device=DEV).to(DEV).eval()
    b = make_batch(2, 24, seed=9, ragged=True)
    first = m(*b[2:])[9].clone()
    m.load_state_dict(fs2_state_dict(cfg, 21))
    second = m(*b[2:])[9].clone()
    fresh = build(cfg, 21).eval()
    want = fresh(*b[2:])[9]
    assert not torch.equal(first, second)
    assert torch.equal(second, want)
    with torch.no_grad():                                  # a direct parameter write after .to() is seen as well
        m.get("mel_linear.bias").add_(1.0)
    third = m(*b[2:])[0]
    assert float((third - fresh(*b[2:])[0]).mean()) > 0.9
