"""GPU: FastSpeech2 on the HIP kernels against (a) outputs of the reference recorded in tests/golden and (b) the
oracle run on the same inputs.

Stated tolerance (bf16 storage / fp32 accumulate vs the reference's fp32; SURVEY.md Appendix A measured the
reference itself under bf16 autocast at rel-RMS 0.4-0.7 %): mel / postnet mel rel-RMS <= 1 %, max-abs <= 0.06;
losses rel 1 %; per-parameter gradient norms rel 6 % (small-norm tensors: abs 2 % of the median norm), individual
gradient tensors rel-RMS 8 % (bf16 activations AND bf16 gradient signals through 10 blocks; 10 % for the
attention query bias, see the comment at the assertion);
LengthRegulator totals and masks exact."""
import copy
import math
import os

import numpy as np
import pytest
import torch

from oracle import fs2 as ofs2
from tests.oracle_util import GOLDEN, fs2_state_dict, rel_rms
from tts_king_amd.synthetic import make_batch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def build(cfg, weight_seed, dropout=True):
    from tts_king_amd.fastspeech2 import FastSpeech2
    m = FastSpeech2(cfg.preprocess_config, cfg.model_config, 65, device=DEV)
    m.load_state_dict(fs2_state_dict(cfg, weight_seed))
    if not dropout:
        m.p_enc = m.p_dec = m.p_var = m.p_post = 0.0
    return m


def check_mel(got, want, name):
    got = got.detach().float().cpu()
    want = torch.as_tensor(want)
    r, a = rel_rms(got, want), float((got - want).abs().max())
    print("%s: rel-RMS %.4f%%  max-abs %.4f" % (name, 100 * r, a))
    assert r <= 0.01 and a <= 0.06, (name, r, a)


def test_eval_teacher_forced_vs_reference_golden(cfg):
    g = np.load(os.path.join(GOLDEN, "fs2_eval_tf.npz"))
    m = build(cfg, int(g["weight_seed"])).eval()
    b = make_batch(int(g["B"]), int(g["L"]), seed=int(g["seed"]), ragged=True)
    o = m(*b[2:])
    torch.cuda.synchronize()
    assert o[8].cpu().tolist() == g["mel_lens"].tolist()
    check_mel(o[0], g["mel"], "mel")
    check_mel(o[9], g["post"], "postnet mel")
    for i, k in ((1, "pitch"), (2, "energy"), (3, "logd")):
        err = float((o[i].cpu() - torch.from_numpy(g[k])).abs().max())
        r = rel_rms(o[i].cpu(), g[k])
        print(k, "max-abs %.4f rel-RMS %.3f%% (rms of the reference %.3f)" % (err, 100 * r, float(np.sqrt((g[k] ** 2).mean()))))
        assert err <= 0.06 and r <= 0.03, (k, err, r)       # predictor outputs are O(0.5): same abs error, larger rel
    src_pad = ofs2.mask_from_lengths(b[4], b[5])
    assert torch.equal(o[5].cpu(), src_pad) and torch.equal(o[6].cpu(), ofs2.mask_from_lengths(b[7], b[8]))
    assert o[0].dtype == torch.float32 and o[10] is None and o[11] is None and len(o) == 12


def test_eval_free_running(cfg):
    g = np.load(os.path.join(GOLDEN, "fs2_eval_free.npz"))
    sd = fs2_state_dict(cfg, int(g["weight_seed"]))
    sd["variance_adaptor.duration_predictor.linear_layer.bias"].fill_(float(g["dur_bias"]))
    from tts_king_amd.fastspeech2 import FastSpeech2
    m = FastSpeech2(cfg.preprocess_config, cfg.model_config, 65, device=DEV).eval()
    m.load_state_dict(sd)
    b = make_batch(int(g["B"]), int(g["L"]), seed=int(g["seed"]), ragged=True)
    dc, pc, ec = [float(x) for x in g["controls"]]
    o = m(b[2], b[3], b[4], b[5], d_control=dc, p_control=pc, e_control=ec)
    d = o[4].cpu()
    want = torch.from_numpy(g["d_rounded"])
    assert d.dtype == torch.float32
    # durations are clamp(round(exp(logd) - 1) * d_control, 0) of a bf16-accurate logd (modules.py:199-204): identical to the reference's
    # except where ITS exp(logd) - 1 sits so close to a rounding boundary n + 0.5 that the logd difference carries it across.  Checked
    # per position, not as a percentage: (1) the HIP log-durations are within the stated 0.06 of the reference's everywhere; (2) the
    # HIP durations are exactly the rounding rule applied to the HIP log-durations; (3) at EVERY position whose duration differs
    # from the golden, the nearest boundary lies between the two pre-rounding values (and so within exp(0.06) - 1 of the
    # reference's, relative to value + 1).
    logd_h, logd_r = o[3].detach().float().cpu(), torch.from_numpy(g["logd"]).float()
    src_mask = torch.arange(d.shape[1])[None, :] < b[4][:, None]
    dl = ((logd_h - logd_r).abs() * src_mask).max()
    assert float(dl) <= 0.06, "log-duration max-abs error %.4f" % float(dl)
    v_h, v_r = torch.exp(logd_h) - 1.0, torch.exp(logd_r) - 1.0
    rule = torch.clamp(torch.round(v_h) * dc, min=0.0)
    off_rule = (rule != d) & src_mask
    # the device's exp may differ from the host's by an ulp: a position where that matters has v_h within 1e-5 of a boundary
    assert bool((((v_h - torch.floor(v_h) - 0.5).abs() < 1e-5) | ~off_rule).all()), "duration_round is not round(exp(logd) - 1) * d_control"
    mism = (d != want) & src_mask
    same = 1.0 - float(mism.sum()) / float(src_mask.sum())
    worst_rel = 0.0
    for bi, li in mism.nonzero().tolist():
        vr, vh = float(v_r[bi, li]), float(v_h[bi, li])
        bnd = math.floor(vr) + 0.5
        if abs(vr - bnd) > 0.5:
            bnd += 1.0
        lo, hi = min(vr, vh), max(vr, vh)
        assert lo - 1e-5 <= bnd <= hi + 1e-5, "duration [%d,%d]: %.4f (HIP) vs %.4f (reference) differ with no rounding boundary between them" % (bi, li, vh, vr)
        assert abs(float(d[bi, li]) - float(want[bi, li])) <= dc + 1e-6
        worst_rel = max(worst_rel, abs(vr - bnd) / (bnd + 1.0))
    print("durations identical: %.1f%% (%d of %d differ, each across a rounding boundary; the reference's value is within %.2e relative of "
          "its boundary at worst); log-duration max-abs error %.4f" % (100 * same, int(mism.sum()), int(src_mask.sum()), worst_rel, float(dl)))
    assert worst_rel <= math.exp(0.06) - 1.0
    assert bool((d[~src_mask] == want[~src_mask]).all())
    di = d.clamp(min=0).trunc().long()
    assert o[8].cpu().tolist() == di.sum(1).tolist()                     # mel_len = sum of truncated durations
    assert o[0].shape[1] == int(di.sum(1).max()) and o[6].shape == (2, o[0].shape[1])
    # same durations as the reference -> same mel: replay the golden durations through the teacher-forced path
    o2 = m(b[2], b[3], b[4], b[5], d_targets=want, max_mel_len=int(g["mel_lens"].max()),
           mel_lens=torch.from_numpy(g["mel_lens"]), p_control=pc, e_control=ec)
    # (pitch/energy embeddings come from predictions*control here as in the golden, since no targets are given)
    # pitch/energy embeddings are picked by bucketize(prediction*control): a bf16-level difference in a prediction
    # that sits next to a bin edge selects the neighbouring embedding row, hence the looser bound on this case
    got = o2[0].detach().float().cpu()
    r = rel_rms(got, g["mel"])
    print("free-running mel (golden durations): rel-RMS %.3f%%" % (100 * r))
    assert r <= 0.03


def test_train_mode_losses_and_gradients(cfg):
    g = np.load(os.path.join(GOLDEN, "fs2_train_p0.npz"))
    from tts_king_amd.loss import FastSpeech2Loss
    m = build(cfg, int(g["weight_seed"]), dropout=False).train()
    loss_fn = FastSpeech2Loss(cfg.preprocess_config, cfg.model_config)
    b = make_batch(int(g["B"]), int(g["L"]), seed=int(g["seed"]), ragged=True)
    o = m(*b[2:])
    ls = loss_fn(b, o)
    assert ls[0].shape == (1,)
    ls[0].backward()                                  # the reference's call site (train.py:44), through the bridge
    torch.cuda.synchronize()
    got = np.array([float(l.sum()) for l in ls])
    print("losses", got, "golden", g["losses"])
    np.testing.assert_allclose(got[:5], g["losses"][:5], rtol=0.01)
    check_mel(o[0], g["mel"], "train mel")
    # train-mode PostNet (Layers.py:133-143) divides every conv output by its BATCH std.  On this synthetic case that makes the
    # PostNet itself an amplifier of whatever error its INPUT carries: the reference PostNet (the oracle, fp32) turns a random
    # perturbation of the mel of the size of the bf16 path's mel error (0.78 % rel-RMS) into 5 % after its first layer (batch std down
    # to 0.06 there) and 15 % at its output (measured on the CPU, fp32 throughout).  So the comparison with the golden output cannot
    # be tight, and a loose bar says nothing about the HIP PostNet.  The sharp statement is two-part:
    #   (1) INPUT error, explained exactly: the oracle PostNet fed with the HIP path's own mel reproduces the HIP PostNet's output up
    #       to the PostNet's own 16-bit storage (five layers of bf16 activations and weights model out at 3.6 % on the CPU): <= 5 %
    #       overall, and per mel channel <= 8 % of that channel's PostNet-output rms;
    #   (2) the end-to-end difference from the reference's recorded output is printed as a diagnostic (the amplification allows ~12 %).
    keep_drop = ofs2._drop
    ofs2._drop = lambda x, p, train: x                      # the oracle's PostNet dropout is hard-coded (0.5): off, as on the HIP side
    try:
        sd_ref = fs2_state_dict(cfg, int(g["weight_seed"]))
        mel_h = o[0].detach().float().cpu()
        with torch.no_grad():
            pn_pred = ofs2.postnet(sd_ref, mel_h, True, None)
    finally:
        ofs2._drop = keep_drop
    post_h, post_r = o[9].detach().float().cpu(), torch.from_numpy(g["post"])
    pn_h = post_h - mel_h
    r_own = rel_rms(pn_h, pn_pred)
    ch_err = (pn_h - pn_pred).pow(2).mean(dim=(0, 1)).sqrt() / pn_pred.pow(2).mean(dim=(0, 1)).sqrt()
    wc = int(ch_err.argmax())
    r = rel_rms(post_h, post_r)
    print("train postnet mel: vs the oracle PostNet on the HIP mel rel-RMS %.3f%% (worst channel %d: %.3f%%); vs the golden %.3f%%"
          % (100 * r_own, wc, 100 * float(ch_err[wc]), 100 * r))
    # THE bars: the HIP PostNet against the reference PostNet on the SAME input
    assert r_own <= 0.05, r_own
    assert float(ch_err.max()) <= 0.08, (wc, float(ch_err[wc]))
    # ... the end-to-end figure against the recorded output (printed above: 8-12 % by build) measures the reference PostNet's own
    # amplification of the 0.78 % mel error (15 % on the CPU for a random perturbation of that size), not this implementation.  Its bar is
    # the measured range plus a margin (0.25 would have let an error of twice the size through); the PostNet's INPUT — the
    # mel_linear path on the window kernel — has its own direct bar against the golden above (check_mel: rel-RMS <= 1 %, max-abs <= 0.06).
    assert r <= 0.14, r
    named = dict(m.named_parameters())
    gn = {str(k): float(v) for k, v in zip(g["grad_keys"], g["grad_norms"])}
    med = float(np.median(list(gn.values())))
    worst = (0.0, None)
    for k, want in gn.items():
        have = float(named[k].grad.norm())
        # the key-projection bias has a TRUE gradient of zero (a constant added to every key's score cancels in the softmax): what
        # the bf16 path leaves there is the rounding noise of dS (its rows sum to zero only before rounding) — bounded at 5 % of
        # the median gradient norm instead of 2 %
        err = abs(have - want) / (want + (0.05 if k.endswith("slf_attn.w_ks.bias") else 0.02) * med)
        if err > worst[0]:
            worst = (err, k, have, want)
    print("worst grad-norm error", worst)
    assert worst[0] <= 0.06, worst
    for name in g.files:
        if name.startswith("grad/"):
            k = name[5:]
            have, want = named[k].grad.detach().cpu(), torch.from_numpy(g[name])
            r = rel_rms(have, want)
            print("grad", k, "rel-RMS %.3f%%" % (100 * r))
            # the query-projection bias gradient is a column sum of dQ = dS K, i.e. of differences P o (dP - sum P dP) that
            # nearly cancel under the almost-uniform attention of random weights: it carries the bf16 rounding of P and dS
            # with the least averaging of all tensors (measured 7.5-8.6 %); every other tensor holds the 8 % bar
            assert r <= (0.10 if k.endswith("slf_attn.w_qs.bias") else 0.08), (k, r)
        if name.startswith("bn/"):
            have = m.state_dict()[name[3:]].cpu()
            np.testing.assert_allclose(have.numpy(), g[name], rtol=2e-2, atol=2e-3)
    for k in g["none_keys"]:
        assert named[str(k)].grad is None
    assert float(named["encoder.src_word_emb.weight"].grad[0].abs().max()) == 0.0    # padding_idx row


def test_train_step_matches_oracle_trainer(cfg):
    """One full main_train_step (fwd, loss, bwd, clip, Adam, LR) vs the oracle trainer on the same batch."""
    from tts_king_amd.loss import FastSpeech2Loss
    from tts_king_amd.optimizer import ScheduledOptim
    from tts_king_amd.train_step import main_train_step, to_device
    c = copy.deepcopy(cfg)
    c.train_config["optimizer"]["grad_acc_step"] = 1
    sd = fs2_state_dict(c, 7)
    m = build(c, 7, dropout=False)
    opt = ScheduledOptim(m, c.train_config, c.model_config, 3999)
    loss_fn = FastSpeech2Loss(c.preprocess_config, c.model_config)
    b = make_batch(2, 48, seed=21, ragged=True)
    vals, out = main_train_step(m, to_device(b, DEV), 1, opt, c, loss_fn)
    mc0 = copy.deepcopy(c.model_config)
    mc0["transformer"]["encoder_dropout"] = mc0["transformer"]["decoder_dropout"] = 0.0
    mc0["variance_predictor"]["dropout"] = 0.0
    tr = ofs2.OracleTrainer(sd, mc0, c.train_config, current_step=3999)
    orig = ofs2._drop
    ofs2._drop = lambda x, p, train: x
    try:
        ovals, _ = tr.train_step(b, 1)
    finally:
        ofs2._drop = orig
    print("losses", vals, ovals)
    np.testing.assert_allclose(vals[:4], ovals[:4], rtol=0.01)
    assert opt.current_step == 4000 and abs(opt.lr() - ofs2.lr_at(4000)) < 1e-12
    # Adam's first step moves every weight by ~lr*sign(g): compare the update direction and size per tensor
    sd0 = fs2_state_dict(c, 7)
    cos_min = 1.0
    for k in tr.keys:
        mine = (m.get(k).detach().cpu() - sd0[k]).flatten().double()
        ref = (tr.sd[k].detach() - sd0[k]).flatten().double()
        if "w_ks.bias" in k or ("postnet" in k and k.endswith("conv.bias")):
            continue    # true gradient is 0 (softmax ignores a key bias; BatchNorm removes a conv bias): Adam amplifies noise
        cos = float((mine @ ref) / (mine.norm() * ref.norm() + 1e-30))
        cos_min = min(cos_min, cos)
        assert abs(float(mine.norm()) / float(ref.norm()) - 1) < 0.1, k
    print("min cosine(update, oracle update) over tensors:", cos_min)
    assert cos_min > 0.9
    assert float(m.flat_buffers()[1].abs().max()) == 0.0                 # zero_grad fused into the step


def test_full_size_step_is_deterministic(cfg):
    """Size-independent property at BASELINE.json's configs[1] size (B=16, L=64, T~423, dropout on): split-K slabs are
    summed in fixed order, the grouped launches carry no atomics and dropout masks are functions of (seed, step, site,
    element) — two runs of three train steps from the same state end in bit-identical parameters and losses."""
    from tts_king_amd.loss import FastSpeech2Loss
    from tts_king_amd.optimizer import ScheduledOptim
    from tts_king_amd.train_step import main_train_step, to_device
    c = copy.deepcopy(cfg)
    c.train_config["optimizer"]["grad_acc_step"] = 1
    b = to_device(make_batch(16, 64, seed=1234), DEV)
    finals = []
    for run in range(2):
        m = build(c, 7, dropout=True)
        opt = ScheduledOptim(m, c.train_config, c.model_config, 0)
        loss_fn = FastSpeech2Loss(c.preprocess_config, c.model_config)
        vals = None
        for s in range(3):
            vals, _ = main_train_step(m, b, s + 1, opt, c, loss_fn)
        finals.append((m.flat_buffers()[0].clone(), [float(v) for v in vals]))
    assert torch.equal(finals[0][0], finals[1][0])
    assert finals[0][1] == finals[1][1]
    assert np.isfinite(finals[0][1]).all()


@pytest.mark.parametrize("B,L,dur_hi", [(1, 3, 3), (3, 7, 40), (2, 130, 9)])
def test_train_step_edge_sizes_vs_oracle_losses(cfg, B, L, dur_hi):
    """Edge sizes of the training batch — one utterance of 3 phonemes / 5 frames (every tile smaller than any kernel's
    block; a single-frame batch is rejected by the reference itself: BatchNorm1d in training mode), ragged lengths, 630
    frames — through the whole step (grouped weight-gradient launches, split-K, deferred column sums): losses against
    the oracle (5 % for the 5-frame batch, whose BatchNorm statistics over 5 rows amplify bf16 rounding; 2 % otherwise),
    finite updated parameters."""
    from tts_king_amd.loss import FastSpeech2Loss
    from tts_king_amd.optimizer import ScheduledOptim
    from tts_king_amd.train_step import main_train_step, to_device
    c = copy.deepcopy(cfg)
    c.train_config["optimizer"]["grad_acc_step"] = 1
    m = build(c, 7, dropout=False)
    opt = ScheduledOptim(m, c.train_config, c.model_config, 100)
    loss_fn = FastSpeech2Loss(c.preprocess_config, c.model_config)
    b = make_batch(B, L, seed=B * 100 + L, ragged=B > 1, dur_hi=dur_hi)
    if int(b[8]) > c.model_config["max_seq_len"]:
        pytest.skip("longer than max_seq_len")
    vals, out = main_train_step(m, to_device(b, DEV), 1, opt, c, loss_fn)
    mc0 = copy.deepcopy(c.model_config)
    mc0["transformer"]["encoder_dropout"] = mc0["transformer"]["decoder_dropout"] = 0.0
    mc0["variance_predictor"]["dropout"] = 0.0
    tr = ofs2.OracleTrainer(fs2_state_dict(c, 7), mc0, c.train_config, current_step=100)
    orig = ofs2._drop
    ofs2._drop = lambda x, p, train: x
    try:
        ovals, _ = tr.train_step(b, 1)
    finally:
        ofs2._drop = orig
    print("B=%d L=%d T=%d losses" % (B, L, int(b[8])), [round(float(v), 4) for v in vals[:5]], [round(float(v), 4) for v in ovals[:5]])
    np.testing.assert_allclose(vals[:4], ovals[:4], rtol=0.05 if int(b[8]) < 16 else 0.02, atol=2e-3)
    assert bool(torch.isfinite(m.flat_buffers()[0]).all())


def test_fused_and_grouped_paths_equal_step_by_step_paths(cfg):
    """The round-2 launch collapses — fc / w_2 + LayerNorm in one kernel (`fused_ln`), the three VariancePredictors and the
    embedding chain as grouped launches (`group_predictors`) — against the step-by-step launches they replace, dropout ON with
    the same counters: same dropout masks, predictions and gradients equal up to bf16 rounding of intermediates."""
    from tts_king_amd import ops
    b = make_batch(4, 40, seed=11, ragged=True)
    res = []
    for fast in (True, False):
        m = build(cfg, 7, dropout=True).train()
        m.fused_ln = m.group_predictors = fast
        dev_b = [t.to(DEV) if torch.is_tensor(t) else t for t in b]
        with torch.no_grad():
            out, ctx = m._forward(True, dev_b[2], dev_b[3], dev_b[4], int(b[5]), dev_b[7], b[8], dev_b[9], dev_b[10], dev_b[11], 1.0, 1.0, 1.0)
            losses, dmel_sum, dpost, dp, de, dd = ops.fs2_loss(out[0], out[8], dev_b[6], dev_b[7], out[1], out[2], out[3], dev_b[11],
                                                               dev_b[9], dev_b[10], dev_b[4], grad_scale=1.0)
            m.backward_native(ctx, dmel_sum, dpost, dp, de, dd)
        torch.cuda.synchronize()
        res.append(([o.float().cpu().clone() for o in (out[0], out[1], out[2], out[3], out[8])], losses.cpu().clone(),
                    {k: p.grad.detach().float().cpu().clone() for k, p in m.named_parameters() if p.grad is not None}))
    (outs_f, loss_f, g_f), (outs_s, loss_s, g_s) = res
    for a, w, name in zip(outs_f, outs_s, ("mel", "pitch", "energy", "logd", "post")):
        r = rel_rms(a, w)
        print("fused vs step-by-step %s rel-RMS %.4f%%" % (name, 100 * r))
        # predictor outputs are O(0.3) sums behind two Dropout(0.5) layers: the one bf16 rounding the fused LayerNorm skips in
        # each of the 8 encoder sub-layers shows up there at ~1.2 %
        # ...; the train-mode PostNet divides by batch statistics (see test_train_mode_losses_and_gradients): 3 % there
        assert r <= {"mel": 0.01, "post": 0.03}.get(name, 0.025), (name, r)
    np.testing.assert_allclose(loss_f[:5].numpy(), loss_s[:5].numpy(), rtol=5e-3)
    worst, worst_pn = (0.0, None), (0.0, None)
    for k in g_s:
        if "w_ks.bias" in k or ("postnet" in k and k.endswith("conv.bias")):
            continue
        r = rel_rms(g_f[k], g_s[k])
        if k.startswith("postnet."):
            if r > worst_pn[0]:
                worst_pn = (r, k)
        elif r > worst[0]:
            worst = (r, k)
    print("fused vs step-by-step worst gradient rel-RMS", worst, "PostNet", worst_pn)
    # the PostNet's train-mode BatchNorm (batch statistics of B*T = 4*~200 rows, dropout 0.5 on) turns the ~1 % difference of its
    # input into a much larger one in its own small parameter gradients (see test_train_mode_losses_and_gradients)
    # (two bf16 paths against each other: each carries its own rounding noise, largest in the sparse embedding-row sums)
    assert worst[0] <= 0.12 and worst_pn[0] <= 0.3, (worst, worst_pn)
