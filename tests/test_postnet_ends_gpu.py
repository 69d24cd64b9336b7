"""GPU: the PostNet's ends on the window kernel (round 5).  reference: fs_two/transformer/Layers.py:85-129,133-143 — convolutions 0 and
4 are Conv1d(80 -> 512, k 5) and Conv1d(512 -> 80, k 5), each followed by BatchNorm1d; fastspeech2.py:104 adds the mel back.  Until
round 4 these four shapes (two convs, two input gradients) ran on the implicit-GEMM tiles with split-K reducers and separate
BatchNorm-statistics launches: 210 us of the step for 12 GFLOP."""
import copy

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def bf(t):
    return t.to(torch.bfloat16)


def _packs(W):
    from tts_king_amd import ops
    cs, k, ds = W.shape
    pk = torch.empty(ops.win_pack_numel(cs, k, ds, False), dtype=torch.bfloat16, device=DEV)
    pkt = torch.empty(ops.win_pack_numel(cs, k, ds, True), dtype=torch.bfloat16, device=DEV)
    ops.win_conv_pack_items([(W, pk, False), (W, pkt, True)])
    return pk, pkt


@pytest.mark.parametrize("B,S", [(16, 423), (2, 64), (3, 65), (1, 7)])
@pytest.mark.parametrize("cin,cout", [(80, 512), (512, 80)])
def test_end_convs_forward_and_input_gradient(B, S, cin, cout):
    """ttsk_win_conv at (80 -> 512) and (512 -> 80), k = 5: forward (fp32 out) and input gradient (bf16 out, transposed pack) against
    fp64 on the same bf16 operands and against the implicit-GEMM path; ragged tile ends (S not a multiple of 64), one-tile utterances."""
    from tts_king_amd import ops
    g = torch.Generator().manual_seed(S * 13 + cin)
    K = 5
    assert ops.win_conv_supported(cin, cout, K) and ops.win_conv_supported(cout, cin, K)
    x = bf(torch.randn(B, S, cin, generator=g)).to(DEV)
    W = bf(torch.randn(cout, K, cin, generator=g) * (cin * K) ** -0.5).to(DEV)          # storage (Cout, k, Cin)
    bias = (0.1 * torch.randn(cout, generator=g)).to(DEV)
    pk, pkt = _packs(W)
    Wt = W.double().cpu().permute(0, 2, 1)
    ref = F.conv1d(x.double().cpu().transpose(1, 2), Wt, bias.double().cpu(), padding=2).transpose(1, 2)
    got = ops.win_conv(x, pk, cout, K, bias=bias, out_dtype=torch.float32)
    assert float((got.double().cpu() - ref).abs().max()) <= 1e-4 * float(ref.abs().max()) + 1e-4
    old = ops.conv1d(x, W, bias, out_dtype=torch.float32)
    assert float((got - old).abs().max()) <= 1e-4 * float(ref.abs().max())
    # input gradient of that conv: dy (B,S,cout) -> dx (B,S,cin)
    dy = bf(torch.randn(B, S, cout, generator=g)).to(DEV)
    dref = F.conv_transpose1d(dy.double().cpu().transpose(1, 2), Wt, padding=2).transpose(1, 2)
    dgot = ops.win_conv(dy, pkt, cin, K)
    assert dgot.dtype == torch.bfloat16 and dgot.shape == (B, S, cin)
    assert float((dgot.double().cpu() - dref).abs().max()) <= 2 ** -8 * float(dref.abs().max()) + 1e-3
    dold = ops.conv1d_dx(dy, W)
    assert float((dgot.float() - dold.float()).abs().max()) <= 2 ** -7 * float(dref.abs().max())
    if cout == 512:
        # this dy -> dx IS conv 0's input gradient (512 -> 80): with the fp32 residual (the mel terms' own gradient) added before the rounding
        R = torch.randn(B, S, cin, generator=g).to(DEV)
        rgot = ops.win_conv_resid(dy, pkt, R, cin, K)
        want = dref + R.double().cpu()
        # ONE rounding of the fp32 sum (round 6; ADVICE r05: the residual used to meet the conv's already-rounded bf16 values): every
        # element within half an ulp of bf16 (8 significant bits: 2^-8 of its own magnitude) of the exact sum
        err = (rgot.double().cpu() - want).abs()
        assert bool((err <= 2 ** -8 * want.abs() + 1e-4).all()), float((err - 2 ** -8 * want.abs()).max())


@pytest.mark.parametrize("B,S,limit", [(16, 423, None), (2, 448, 423), (3, 70, 61), (1, 64, None)])
@pytest.mark.parametrize("cin,cout", [(80, 512), (512, 80)])
def test_end_convs_emit_batchnorm_partials(B, S, limit, cin, cout):
    """ttsk_win_conv_stats at the two end shapes: output bit-identical to ttsk_win_conv, statistics partials that give ttsk_bn_train_apply
    the mean / rstd ttsk_bn_stats_slab computes from the stored rows (frame limit included)."""
    from tts_king_amd import ops
    g = torch.Generator().manual_seed(B * 7 + S + cin)
    x = bf(torch.randn(B, S, cin, generator=g)).to(DEV)
    W = bf(torch.randn(cout, 5, cin, generator=g) * (5 * cin) ** -0.5).to(DEV)
    bias = (0.1 * torch.randn(cout, generator=g)).to(DEV)
    pk, _ = _packs(W)
    fl = None if limit is None else (torch.tensor([limit], dtype=torch.int32, device=DEV), S)
    want = ops.win_conv(x, pk, cout, 5, bias=bias, out_dtype=torch.float32)
    got, stats = ops.win_conv_stats(x, pk, cout, 5, bias=bias, frame_limit=fl)
    assert torch.equal(got, want) and stats.shape == (B * ((S + 63) // 64), 2 * cout)
    gamma, beta = (1 + 0.1 * torch.randn(cout, generator=g)).to(DEV), (0.1 * torch.randn(cout, generator=g)).to(DEV)
    rows = B * S
    z = lambda: (torch.zeros(cout, device=DEV), torch.ones(cout, device=DEV), torch.zeros(1, dtype=torch.int64, device=DEV))
    o0, m0, r0 = ops.bn_train(want.view(rows, cout), *z(), gamma, beta, True, frame_limit=fl)
    o1, m1, r1 = ops.bn_train(got.view(rows, cout), *z(), gamma, beta, True, frame_limit=fl, partials=stats)
    np.testing.assert_allclose(m1.cpu().numpy(), m0.cpu().numpy(), rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(r1.cpu().numpy(), r0.cpu().numpy(), rtol=1e-5, atol=1e-6)
    assert float((o1.float() - o0.float()).abs().max()) <= 2 ** -7


@pytest.mark.parametrize("B,S,limit,p", [(16, 423, None, 0.5), (2, 448, 423, 0.5), (3, 70, 61, 0.0)])
def test_last_conv_input_gradient_emits_batchnorm_backward_partials(B, S, limit, p):
    """ttsk_win_conv_bnb at (80 -> 512): the PostNet's LAST conv's input gradient, which is the upstream gradient of layer 3's
    BatchNorm: output bit-identical to ttsk_win_conv, and the partials give ttsk_bn_bwd_apply_slab the sums ttsk_bn_bwd_stats_slab
    computes from the stored gradient (tanh, keep bits, frame limit)."""
    from tts_king_amd import ops
    g = torch.Generator().manual_seed(B * 11 + S)
    C, Cm = 512, 80
    dy = bf(torch.randn(B, S, Cm, generator=g)).to(DEV)
    W = bf(torch.randn(Cm, 5, C, generator=g) * (5 * C) ** -0.5).to(DEV)                 # conv 4: storage (80, 5, 512)
    _, pkt = _packs(W)
    fl = None if limit is None else (torch.tensor([limit], dtype=torch.int32, device=DEV), S)
    rows = B * S
    yc = torch.randn(rows, C, generator=g).to(DEV)
    gamma, beta = (1 + 0.1 * torch.randn(C, generator=g)).to(DEV), (0.1 * torch.randn(C, generator=g)).to(DEV)
    rng = ops.rng_of(ops.optim_state(DEV, seed=1234)) if p > 0 else None
    z = lambda: (torch.zeros(C, device=DEV), torch.ones(C, device=DEV), torch.zeros(1, dtype=torch.int64, device=DEV))
    _, mean, rstd, keep = ops.bn_train(yc, *z(), gamma, beta, True, p=p, site=303, rng=rng, frame_limit=fl, want_keep=True)
    want = ops.win_conv(dy, pkt, C, 5)
    got, stats = ops.win_conv_bnb(dy, pkt, C, 5, yc, mean, rstd, gamma, beta, True, p=p, keep=keep, frame_limit=fl)
    assert torch.equal(got, want) and stats.shape == (B * ((S + 63) // 64), 2 * C)
    outs = []
    for part in (None, stats):
        dg, db = torch.zeros(C, device=DEV), torch.zeros(C, device=DEV)
        dx = ops.bn_bwd(want.view(rows, C), yc, mean, rstd, gamma, beta, True, p=p, site=303, rng=rng, dgamma=dg, dbeta=db, frame_limit=fl,
                        keep=keep, accumulate=False, partials=part)
        outs.append((dx.float().cpu(), dg.cpu(), db.cpu()))
    (dx0, dg0, db0), (dx1, dg1, db1) = outs
    sc = float(dg0.abs().max()) + float(db0.abs().max())
    np.testing.assert_allclose(dg1.numpy(), dg0.numpy(), rtol=2e-5, atol=2e-5 * sc)
    np.testing.assert_allclose(db1.numpy(), db0.numpy(), rtol=2e-5, atol=2e-5 * sc)
    assert float((dx1 - dx0).abs().max()) <= 2 ** -7 * float(dx0.abs().max())


def test_step_with_the_ends_on_the_window_kernel_matches_the_gemm_path(cfg):
    """One full training step (dropout off, ragged lengths) with `postnet_ends_win` on and off: same losses to fp32 rounding of different
    summation orders, gradients of every PostNet / mel_linear / decoder parameter group within 1e-3 of their norm, updated packs fresh
    (a second step agrees as well: the odd packs are rewritten behind the optimizer step)."""
    from tts_king_amd.fastspeech2 import FastSpeech2
    from tts_king_amd.graph import make_enqueue
    from tts_king_amd.loss import FastSpeech2Loss
    from tts_king_amd.optimizer import ScheduledOptim
    from tts_king_amd.synthetic import make_batch
    from tts_king_amd.train_step import to_device
    c = copy.deepcopy(cfg)
    c.train_config["optimizer"]["grad_acc_step"] = 1
    batch = to_device(make_batch(5, 48, seed=31, ragged=True), DEV)
    res = []
    for on in (False, True):
        m = FastSpeech2(c.preprocess_config, c.model_config, 65, device=DEV, seed=7)
        m.postnet_ends_win = on
        m._build_packs()
        m.sync_shadow(force=True)
        m.p_enc = m.p_dec = m.p_var = m.p_post = 0.0
        m.train()
        assert (("pn", "postnet.convolutions.0.0.conv.weight") in m._w1_packed) == on
        opt = ScheduledOptim(m, c.train_config, c.model_config, 0)
        enq = make_enqueue(m, opt, c, FastSpeech2Loss(c.preprocess_config, c.model_config))
        l1, _ = enq(batch)
        g1 = m.flat_buffers()[1].cpu().clone()
        l2, _ = enq(batch)
        torch.cuda.synchronize()
        res.append((l1.cpu().clone(), l2.cpu().clone(), g1, m))
    (a1, a2, ga, ma), (b1, b2, gb, mb) = res
    np.testing.assert_allclose(b1.numpy(), a1.numpy(), rtol=2e-3)
    np.testing.assert_allclose(b2.numpy(), a2.numpy(), rtol=5e-3)
    for name, lo in ma.group_offsets().items():
        hi = min([o for o in ma.group_offsets().values() if o > lo] + [ga.numel()])
        na = float(ga[lo:hi].norm())
        assert float((gb[lo:hi] - ga[lo:hi]).norm()) <= 2e-2 * na + 1e-6, name


@pytest.mark.parametrize("B,S", [(16, 423), (3, 65), (1, 7)])
def test_mel_linear_on_the_window_kernel(B, S):
    """mel_linear (fastspeech2.py:102, Linear(256 -> 80)) as a k = 1 conv on the five-wave instance with an fp32 output and its bf16 copy
    (ttsk_win_conv_dual), and its input gradient (80 -> 256 on the padded transposed pack), against fp64 and the GEMM path."""
    from tts_king_amd import ops
    g = torch.Generator().manual_seed(S)
    x = bf(torch.randn(B, S, 256, generator=g)).to(DEV)
    W = bf(torch.randn(80, 1, 256, generator=g) / 16).to(DEV)
    bias = (0.1 * torch.randn(80, generator=g)).to(DEV)
    pk, pkt = _packs(W)
    ref = x.double().cpu().view(-1, 256) @ W.double().cpu().view(80, 256).t() + bias.double().cpu()
    out, out16 = ops.win_conv_dual(x, pk, 80, 1, bias=bias)
    assert out.dtype == torch.float32 and out16.dtype == torch.bfloat16
    assert float((out.double().cpu().view(-1, 80) - ref).abs().max()) <= 1e-4 * float(ref.abs().max()) + 1e-5
    assert torch.equal(out16, out.to(torch.bfloat16))
    old = ops.linear(x.view(-1, 256), W.view(80, 256), bias, out_dtype=torch.float32)
    assert float((out.view(-1, 80) - old).abs().max()) <= 1e-4 * float(ref.abs().max())
    dy = bf(torch.randn(B, S, 80, generator=g)).to(DEV)
    dref = dy.double().cpu().view(-1, 80) @ W.double().cpu().view(80, 256)
    dx = ops.win_conv(dy, pkt, 256, 1)
    assert float((dx.double().cpu().view(-1, 256) - dref).abs().max()) <= 2 ** -8 * float(dref.abs().max()) + 1e-3
    dold = ops.linear_dx(dy.view(-1, 80), W.view(80, 256))
    assert float((dx.view(-1, 256).float() - dold.float()).abs().max()) <= 2 ** -7 * float(dref.abs().max())


def test_traced_step_accounts_every_family(cfg):
    """bench.py's roofline leg runs a step with ops.GEMM_TRACE set: every launch wrapper's FLOP formula is then evaluated on the
    wrapper's own arguments.  (Round 5: a wrapper with a new signature behind the shared formula failed there, and only there.)"""
    from tts_king_amd import ops
    from tts_king_amd.fastspeech2 import FastSpeech2
    from tts_king_amd.graph import make_enqueue
    from tts_king_amd.loss import FastSpeech2Loss
    from tts_king_amd.optimizer import ScheduledOptim
    from tts_king_amd.synthetic import make_batch
    from tts_king_amd.train_step import to_device
    c = copy.deepcopy(cfg)
    c.train_config["optimizer"]["grad_acc_step"] = 1
    batch = to_device(make_batch(3, 40, seed=5, ragged=True), DEV)
    m = FastSpeech2(c.preprocess_config, c.model_config, 65, device=DEV, seed=3).train()
    opt = ScheduledOptim(m, c.train_config, c.model_config, 0)
    enq = make_enqueue(m, opt, c, FastSpeech2Loss(c.preprocess_config, c.model_config))
    enq(batch)
    ops.GEMM_TRACE = []
    try:
        enq(batch)
        torch.cuda.synchronize()
        trace = ops.GEMM_TRACE
    finally:
        ops.GEMM_TRACE = None
    kinds = {t[3] for t in trace}
    assert "win_conv" in kinds and len(trace) > 50
    assert all(t[2] >= 0.0 for t in trace)
