"""GPU: flash attention (tts_king_amd/csrc/flash_attn.hip) against fp64 math on the same bf16-rounded inputs, and against the
general scores-GEMM + softmax path inside a train step.
reference: fs_two/transformer/Modules.py:14-24, SubLayers.py:44-60.  Tolerances: O is rounded to bf16 once (2^-8 relative), the
gradients additionally see the bf16 rounding of P and dS (stated at the assertions).  Ragged key lengths, a sequence shorter
than one tile, one that is not a tile multiple and one past max_seq_len are covered."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
BF = torch.bfloat16


def ref_attention(qkv, lens, B, H, S):
    d = qkv.shape[1] // 3
    dk = d // H
    x = qkv.double().view(B, S, 3, H, dk)
    q, k, v = x[:, :, 0].permute(0, 2, 1, 3), x[:, :, 1].permute(0, 2, 1, 3), x[:, :, 2].permute(0, 2, 1, 3)   # (B,H,S,dk)
    s = q @ k.transpose(-1, -2) / dk ** 0.5
    mask = torch.arange(S)[None, :] >= lens[:, None]
    s = s.masked_fill(mask[:, None, None, :], float("-inf"))
    p = torch.softmax(s, dim=-1)
    o = (p @ v).permute(0, 2, 1, 3).reshape(B * S, d)
    return p.reshape(B * H, S, S), o, (q, k, v)


def test_flash_and_gemm_softmax_paths_agree(cfg):
    """A train step's forward + backward with flash attention vs the general path (scores GEMM + masked softmax + P V GEMM: what
    head sizes other than 128 run), same weights, dropout off."""
    from tts_king_amd.fastspeech2 import FastSpeech2
    from tts_king_amd.loss import FastSpeech2Loss
    from tts_king_amd.synthetic import make_batch
    from tests.oracle_util import fs2_state_dict
    outs = []
    for fused in (True, False):
        m = FastSpeech2(cfg.preprocess_config, cfg.model_config, 65, device=DEV)
        m.load_state_dict(fs2_state_dict(cfg, 7))
        m.p_enc = m.p_dec = m.p_var = m.p_post = 0.0
        m.flash_attention = fused
        m.train()
        b = make_batch(2, 40, seed=3, ragged=True)
        o = m(*b[2:])
        ls = FastSpeech2Loss(cfg.preprocess_config, cfg.model_config)(b, o)
        ls[0].backward()
        torch.cuda.synchronize()
        outs.append((o[0].detach().float().cpu(), m.flat_buffers()[1].clone().cpu()))
    (mel_a, g_a), (mel_b, g_b) = outs
    assert float((mel_a - mel_b).abs().max()) <= 0.05
    rel = float((g_a - g_b).norm() / g_b.norm())
    print("flash vs GEMM + softmax attention: grad rel diff %.4f" % rel)
    assert rel <= 0.02


@pytest.mark.parametrize("B,H,S", [(3, 2, 64), (2, 2, 423), (2, 2, 37), (1, 2, 200), (2, 1, 130), (1, 2, 1000)])
def test_flash_attention_fwd_bwd(B, H, S):
    """ttsk_flash_attention_fwd / _bwd (no S x S tensor in HBM) against fp64 math: O, LSE, dQ, dK, dV, ragged key lengths,
    a length beyond S (train-mode truncation), random dO on every query row."""
    from tts_king_amd import ops
    g = torch.Generator().manual_seed(S + B)
    d = H * 128
    qkv = (torch.randn(B * S, 3 * d, generator=g) * 0.7).to(BF)
    lens = torch.randint(max(1, S // 2), S + 1, (B,), generator=g)
    lens[0] = S + 232 if S == 1000 else S
    lens_c = lens.clamp(max=S)
    P, O, (q, k, v) = ref_attention(qkv.float(), lens_c, B, H, S)
    o, lse, o32 = ops.flash_attention_fwd(qkv.to(DEV), lens.to(DEV), B, H, S, want_lse=True)
    assert float((o.float().cpu().double() - O).abs().max()) <= 0.02 * float(O.abs().max())
    assert torch.equal(o32.to(BF), o)                       # the fp32 copy the backward's delta is taken from
    sc = (q @ k.transpose(-1, -2)) / 128 ** 0.5
    mask = torch.arange(S)[None, :] >= lens_c[:, None]
    want_lse = torch.logsumexp(sc.masked_fill(mask[:, None, None, :], float("-inf")), dim=-1).reshape(B * H, S)
    assert float((lse.cpu().double() - want_lse).abs().max()) <= 2e-3
    o2, none, none32 = ops.flash_attention_fwd(qkv.to(DEV), lens.to(DEV), B, H, S, want_lse=False)
    assert none is None and none32 is None and torch.equal(o2, o)
    # ---- backward: autograd of the fp64 reference through O
    do = torch.randn(B * S, d, generator=g).to(BF)
    x = qkv.double().clone().requires_grad_(True)
    xv = x.view(B, S, 3, H, 128)
    qq, kk, vv = xv[:, :, 0].permute(0, 2, 1, 3), xv[:, :, 1].permute(0, 2, 1, 3), xv[:, :, 2].permute(0, 2, 1, 3)
    pp = torch.softmax(((qq @ kk.transpose(-1, -2)) / 128 ** 0.5).masked_fill(mask[:, None, None, :], float("-inf")), dim=-1)
    oo = (pp @ vv).permute(0, 2, 1, 3).reshape(B * S, d)
    oo.backward(do.double())
    dqkv = ops.flash_attention_bwd(qkv.to(DEV), o, do.to(DEV), lse, lens.to(DEV), B, H, S, o32=o32).float().cpu().double()
    dqkv_b = ops.flash_attention_bwd(qkv.to(DEV), o, do.to(DEV), lse, lens.to(DEV), B, H, S).float().cpu().double()   # delta from bf16 O
    for name, lo in (("dQ", 0), ("dK", d), ("dV", 2 * d)):
        got, want = dqkv[:, lo:lo + d], x.grad[:, lo:lo + d]
        err = float((got - want).abs().max())
        err_b = float((dqkv_b[:, lo:lo + d] - want).abs().max())
        print("%s S=%d max-abs err %.4g (delta from bf16 O: %.4g) of %.4g" % (name, S, err, err_b, float(want.abs().max())))
        assert err <= 0.03 * float(want.abs().max()), (name, err)
        assert err_b <= 0.04 * float(want.abs().max()), (name, err_b)
    # keys past the utterance: exactly zero dK / dV
    for bi in range(B):
        L_ = int(lens_c[bi])
        if L_ < S:
            assert float(dqkv[bi * S + L_:(bi + 1) * S, d:].abs().max()) == 0.0
