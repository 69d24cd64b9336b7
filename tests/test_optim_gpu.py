"""GPU: the optimizer step's two forms — `ttsk_optim_step` + `ttsk_win_conv_pack_table` (round 2) and `ttsk_optim_step_packed`, whose
Adam launch writes the window kernels' weight packs itself — leave bit-identical parameters, moments, bf16 shadow and packs; the
global gradient norm the step records equals the norm of the flat gradient buffer.
reference: train.py:47-54 (clip -> lr -> Adam -> zero_grad), torch.optim.Adam."""
import copy

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _model(cfg, seed):
    from tts_king_amd.fastspeech2 import FastSpeech2
    from tts_king_amd.optimizer import ScheduledOptim
    c = copy.deepcopy(cfg)
    m = FastSpeech2(c.preprocess_config, c.model_config, 65, device=DEV, seed=seed).train()
    m.sync_shadow()
    opt = ScheduledOptim(m, c.train_config, c.model_config, 0)
    return m, opt


@pytest.mark.parametrize("scale", [1e-3, 3.0])          # below and above the clip threshold (max_norm 1.0)
def test_packed_adam_equals_adam_then_pack(cfg, scale):
    res = []
    for packed in (False, True):
        m, opt = _model(cfg, 3)
        assert m._adam_tables is not None, "the shipped configuration's packed weights all tile"
        if not packed:
            m._adam_tables = None
        g = torch.Generator(device=DEV).manual_seed(11)
        flat, grad, shadow = m.flat_buffers()
        for step in range(3):
            grad.copy_(torch.randn(grad.shape, generator=g, device=DEV) * scale * (1 + step) / grad.numel() ** 0.5)
            want_norm = float(grad.double().norm())
            opt.step_and_update_lr()
            torch.cuda.synchronize()
            assert abs(opt.grad_norm() - want_norm) <= 1e-5 * want_norm
            assert float(grad.abs().max()) == 0.0
        packs = torch.cat([v.view(-1) for _, v in sorted(m._w1_packed.items())])
        res.append((flat.clone(), opt.exp_avg.clone(), opt.exp_avg_sq.clone(), shadow.clone(), packs.clone(), opt.current_step))
    for a, b, name in zip(res[0], res[1], ("params", "exp_avg", "exp_avg_sq", "shadow", "packs")):
        assert torch.equal(a, b), name
    assert res[0][5] == res[1][5] == 3
    # the packs are the re-layout of the shadow: a fresh pack launch from the shadow changes nothing
    m.refresh_packed()
    torch.cuda.synchronize()
    packs2 = torch.cat([v.view(-1) for _, v in sorted(m._w1_packed.items())])
    assert torch.equal(packs2, res[1][4])


def test_overwrite_backward_equals_zero_then_accumulate(cfg):
    """The sync-free step does not zero the gradient buffer after Adam: the next backward OVERWRITES it (backward_native(accumulate=
    False)).  Every producer of a gradient must then write every element it owns: a backward into a buffer poisoned with 1e30 must
    give, through every parameter's `.grad` view, bit-for-bit what a backward into a zeroed buffer with accumulate=True gives (and
    the poison must be gone everywhere but in the alignment padding between parameters, which nothing reads)."""
    from tests.oracle_util import fs2_state_dict
    from tts_king_amd import ops
    from tts_king_amd.fastspeech2 import FastSpeech2
    from tts_king_amd.synthetic import make_batch
    c = copy.deepcopy(cfg)
    b = make_batch(3, 40, seed=21, ragged=True)
    res = []
    for poison in (False, True):
        m = FastSpeech2(c.preprocess_config, c.model_config, 65, device=DEV).train()
        m.load_state_dict(fs2_state_dict(c, 7))
        m.p_enc = m.p_dec = m.p_var = m.p_post = 0.0
        grad = m.flat_buffers()[1]
        grad.fill_(1e30 if poison else 0.0)
        dev_b = [t.to(DEV) if torch.is_tensor(t) else t for t in b]
        with torch.no_grad():
            out, ctx = m._forward(True, dev_b[2], dev_b[3], dev_b[4], int(b[5]), dev_b[7], b[8], dev_b[9], dev_b[10], dev_b[11], 1.0, 1.0, 1.0)
            _, dmel_sum, dpost, dp, de, dd = ops.fs2_loss(out[0], out[8], dev_b[6], dev_b[7], out[1], out[2], out[3], dev_b[11],
                                                          dev_b[9], dev_b[10], dev_b[4], grad_scale=1.0)
            m.backward_native(ctx, dmel_sum, dpost, dp, de, dd, accumulate=not poison)
        torch.cuda.synchronize()
        assert m.grads_partial
        res.append({k: p.grad.detach().clone() for k, p in m.named_parameters() if p.grad is not None})
    assert set(res[0]) == set(res[1]) and len(res[0]) > 200
    for k in res[0]:
        assert float(res[1][k].abs().max()) < 1e29, "%s: stale values survive an overwriting backward" % k
        assert torch.equal(res[0][k], res[1][k]), k


def test_dwconv_path_equals_grouped_gemm_path(cfg):
    """FastSpeech2.dwconv (w_1's weight gradient on csrc/dwconv.hip, the other 256-multiple weight gradients on csrc/dwgemm.hip, rows
    past each utterance's length skipped) against the grouped GEMM path on the same step: the weights those kernels produce within fp32
    summation-order noise, every other gradient bit-identical."""
    from tests.oracle_util import fs2_state_dict
    from tts_king_amd import ops
    from tts_king_amd.fastspeech2 import FastSpeech2
    from tts_king_amd.synthetic import make_batch
    c = copy.deepcopy(cfg)
    b = make_batch(5, 48, seed=33, ragged=True)
    res = []
    for use in (False, True):
        m = FastSpeech2(c.preprocess_config, c.model_config, 65, device=DEV).train()
        m.load_state_dict(fs2_state_dict(c, 7))
        m.dwconv = use
        dev_b = [t.to(DEV) if torch.is_tensor(t) else t for t in b]
        counts = {}
        ops.LAUNCH_COUNTS = counts
        try:
            with torch.no_grad():
                out, ctx = m._forward(True, dev_b[2], dev_b[3], dev_b[4], int(b[5]), dev_b[7], b[8], dev_b[9], dev_b[10], dev_b[11], 1.0, 1.0, 1.0)
                _, dmel_sum, dpost, dp, de, dd = ops.fs2_loss(out[0], out[8], dev_b[6], dev_b[7], out[1], out[2], out[3], dev_b[11],
                                                              dev_b[9], dev_b[10], dev_b[4], grad_scale=1.0)
                m.backward_native(ctx, dmel_sum, dpost, dp, de, dd)
            torch.cuda.synchronize()
        finally:
            ops.LAUNCH_COUNTS = None
        assert (counts.get("dwconv", 0) > 0) == use, counts
        res.append({k: p.grad.detach().clone() for k, p in m.named_parameters() if p.grad is not None})
    worst = 0.0
    for k in res[0]:
        if k.endswith(("pos_ffn.w_1.weight", "pos_ffn.w_2.weight", "slf_attn.fc.weight", "slf_attn.w_qs.weight", "slf_attn.w_ks.weight",
                       "slf_attn.w_vs.weight", "conv1d_1.conv.weight", "conv1d_2.conv.weight")) or (k.startswith("postnet.convolutions.") and k.endswith("0.conv.weight")):
            scale = float(res[0][k].abs().max())
            err = float((res[0][k] - res[1][k]).abs().max()) / scale
            worst = max(worst, err)
            assert err <= 1e-5, (k, err)
        else:
            assert torch.equal(res[0][k], res[1][k]), k
    print("weight gradients, dwconv / dwgemm vs grouped GEMM: max difference %.2e of max |g|" % worst)
