"""GPU: HiFi-GAN generator on the HIP kernels against (a) the waveform the reference produced (tests/golden/
hifi_b2_t32.npz) and (b) the oracle on the same inputs, through the reference's surface (weight-normed
state_dict -> remove_weight_norm -> forward; HIFIapi.generate -> int16).

Stated tolerance (fp16 activations/weights — the generator is inference-only, so the product path stores fp16, which
has 3 more mantissa bits than bf16 at the same MFMA rate — fp32 accumulate, vs the reference's fp32): waveform
rel-RMS <= 0.5 %, max-abs <= 0.01 of full scale; int16 samples within 0.5 % of full scale of the reference's;
weight-norm folding fp32-exact (rtol 1e-5 per element); the int16 cast of an identical float input: bit-exact.
(With bf16 storage the same path measures 0.7-1.1 % rel-RMS: 0.58 % from the weights' and 0.58 % from the MFMA
operands' 8-bit mantissas alone — `test_bf16_storage_variant` keeps that variant under a 1.5 % bar.)"""
import copy
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import hifigan as ohifi
from tests.oracle_util import GOLDEN, hifi_state_dict_wn, rel_rms
from tts_king_amd.synthetic import make_mel

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def build(cfg, weight_seed, fold_on_device=True):
    from tts_king_amd.hifigan import Generator
    g = Generator(cfg.hifi)
    g.load_state_dict(hifi_state_dict_wn(weight_seed))
    if fold_on_device:
        g.to(DEV)
        g.remove_weight_norm()
    else:
        g.remove_weight_norm()
        g.to(DEV)
    return g.eval()


def test_state_dict_keys_and_fold(cfg):
    g = np.load(os.path.join(GOLDEN, "hifi_b2_t32.npz"))
    spec = np.load(os.path.join(GOLDEN, "hifi_state_dict_spec.npz"))
    from tts_king_amd.hifigan import Generator
    gen = Generator(cfg.hifi)
    assert list(gen.state_dict().keys()) == [str(k) for k in spec["wn_keys"]]
    gen.load_state_dict(hifi_state_dict_wn(int(g["weight_seed"])))
    gen.to(DEV)
    gen.remove_weight_norm()
    sd = gen.state_dict()
    assert sorted(sd.keys()) == sorted(str(k) for k in spec["keys"]) and len(sd) == int(g["n_folded_keys"])
    for name in g.files:
        if name.startswith("fold/"):
            np.testing.assert_allclose(sd[name[5:]].cpu().reshape(-1)[:64].numpy(), g[name], rtol=1e-5, atol=1e-7)
        if name.startswith("foldnorm/"):
            # the golden norm itself was accumulated in fp32 by torch (3e-5 off the fp64 value for the largest tensor)
            np.testing.assert_allclose(float(sd[name[9:]].double().norm()), float(g[name]), rtol=1e-4)


@pytest.mark.parametrize("fold_on_device", [True, False])
def test_waveform_vs_reference_golden(cfg, fold_on_device):
    g = np.load(os.path.join(GOLDEN, "hifi_b2_t32.npz"))
    gen = build(cfg, int(g["weight_seed"]), fold_on_device)
    mel = make_mel(int(g["B"]), int(g["T"]), seed=int(g["seed"]))
    wav = gen(mel.to(DEV))
    torch.cuda.synchronize()
    assert wav.shape == (2, 1, 8192) and wav.dtype == torch.float32
    r, a = rel_rms(wav.cpu(), g["wav"]), float((wav.cpu() - torch.from_numpy(g["wav"])).abs().max())
    print("waveform vs reference: rel-RMS %.3f%%  max-abs %.5f (rms of the reference %.4f)" % (100 * r, a, float(np.sqrt((g["wav"] ** 2).mean()))))
    assert r <= 0.005 and a <= 0.01


def test_bf16_storage_variant(cfg):
    from tts_king_amd.ops import bf16
    g = np.load(os.path.join(GOLDEN, "hifi_b2_t32.npz"))
    gen = build(cfg, int(g["weight_seed"]))
    gen.act_dtype = bf16
    wav = gen(make_mel(int(g["B"]), int(g["T"]), seed=int(g["seed"])).to(DEV)).cpu()
    r = rel_rms(wav, g["wav"])
    print("bf16 storage variant vs reference: rel-RMS %.3f%%" % (100 * r))
    assert r <= 0.015


def test_hifiapi_generate_int16(cfg):
    """HIFIapi: random-init from the seed when weights_path is null, int16 output, use_cpu rejected."""
    import hifiapi
    g = np.load(os.path.join(GOLDEN, "hifi_b2_t32.npz"))
    c = copy.deepcopy(cfg)
    api = hifiapi.HIFIapi(c, "cuda:0")
    api.model.load_state_dict({k: v for k, v in ohifi.fold_weight_norm(hifi_state_dict_wn(int(g["weight_seed"]))).items()})
    mel = make_mel(int(g["B"]), int(g["T"]), seed=int(g["seed"]))
    i16 = api.generate(mel)
    assert i16.dtype == np.int16 and i16.shape == (2, 1, 8192)
    assert np.abs(i16.astype(np.int32) - g["int16"].astype(np.int32)).max() <= 164          # 0.5 % of full scale
    # the cast itself is exact: same float input -> same int16 as numpy astype (truncation toward zero)
    wav = api(mel)
    assert np.array_equal(ohifi.to_int16(wav.cpu(), c.hifi.MAX_WAV_VALUE), api.generate(mel))
    with pytest.raises(NotImplementedError):
        api.train()
    c.model_config["vocoder"]["use_cpu"] = True
    from tts_king_amd.lib import TtskError
    with pytest.raises(TtskError):
        hifiapi.HIFIapi(c, "cuda:0")


@pytest.mark.parametrize("B,T", [(1, 1), (3, 7), (1, 100), (2, 384), (8, 384)])
def test_waveform_vs_oracle_shapes(cfg, B, T):
    """Ragged / tiny time lengths and the full BASELINE.json configs[2] batch (B=8, T=384: 8 x 98,304 samples, a few seconds of CPU
    oracle) against the oracle on the same seeded weights."""
    sdw = hifi_state_dict_wn(11)
    gen = build(cfg, 11)
    mel = make_mel(B, T, seed=100 + T)
    with torch.no_grad():
        want = ohifi.generator(ohifi.fold_weight_norm(sdw), cfg.hifi, mel)
    got = gen(mel.to(DEV)).cpu()
    assert got.shape == want.shape == (B, 1, 256 * T)
    r = rel_rms(got, want)
    print("B=%d T=%d rel-RMS %.3f%% max-abs %.5f" % (B, T, 100 * r, float((got - want).abs().max())))
    assert r <= 0.005 and float((got - want).abs().max()) <= 0.01


def test_batch_independence_and_determinism(cfg):
    """Size-independent properties at the BASELINE config (B=8, T=384): utterances do not interact, replays are
    bit-identical, and every sample is inside tanh's range."""
    gen = build(cfg, 3)
    mel = make_mel(8, 384, seed=1234).to(DEV)
    w1 = gen(mel)
    w2 = gen(mel)
    assert torch.equal(w1, w2)
    assert w1.shape == (8, 1, 98304) and float(w1.abs().max()) <= 1.0 and bool(torch.isfinite(w1).all())
    single = gen(mel[5:6])
    # same utterance alone: the split-K plan may differ with the row count, so only fp32 summation order changes
    assert rel_rms(single.cpu(), w1[5:6].cpu()) <= 1e-3


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("C,K,B,ln", [(32, 3, 2, 700), (32, 7, 1, 256), (32, 11, 2, 1000), (64, 3, 1, 130), (64, 7, 2, 515),
                                      (64, 11, 3, 300), (32, 11, 1, 5), (64, 11, 1, 1)])
def test_fused_resblock1_vs_oracle(C, K, B, ln, dtype):
    """The fused six-conv ResBlock1 kernel against the oracle's res_block1 on the same bf16-rounded input and
    weights: only fp32 summation order and the bf16 rounding of the two LDS-resident intermediates differ.
    Lengths that are not tile multiples, shorter than the halo, and a single frame are covered; mode 1/2 = MRF sum."""
    from tts_king_amd import ops
    g = torch.Generator().manual_seed(C * K + ln)
    bf = lambda t: t.to(dtype)
    x = bf(torch.randn(B, ln, C, generator=g))
    ws = [bf(torch.randn(C, C, K, generator=g) * (C * K) ** -0.5) for _ in range(6)]
    bs = [0.1 * torch.randn(C, generator=g) for _ in range(6)]
    sd = {}
    for m in range(3):
        sd["r.convs1.%d.weight" % m], sd["r.convs1.%d.bias" % m] = ws[2 * m].float(), bs[2 * m]
        sd["r.convs2.%d.weight" % m], sd["r.convs2.%d.bias" % m] = ws[2 * m + 1].float(), bs[2 * m + 1]
    with torch.no_grad():
        want = ohifi.res_block1(sd, "r.", x.float().transpose(1, 2), K, (1, 3, 5)).transpose(1, 2)
    wk = [ops.pack_resblock_weight(w.float().to(DEV), dtype=dtype) for w in ws]      # fragment-major packs
    bd = [b.to(DEV) for b in bs]
    out = torch.full((B, ln, C), 7.0, dtype=dtype, device=DEV)
    ops.hifi_resblock1(x.to(DEV), wk, bd, (1, 3, 5), out, K, mode=0)
    got = out.float().cpu()
    r = rel_rms(got, want)
    print("C=%d K=%d len=%d rel-RMS %.3f%% max-abs %.4f" % (C, K, ln, 100 * r, float((got - want).abs().max())))
    tol = 0.008 if dtype == torch.bfloat16 else 0.001
    assert r <= tol and float((got - want).abs().max()) <= 6 * tol * float(want.abs().max())
    # MRF accumulation modes: out2 = (out + y) / 3 computed in fp32 from the bf16 values, rounded once
    out2 = out.clone()
    ops.hifi_resblock1(x.to(DEV), wk, bd, (1, 3, 5), out2, K, mode=1)
    assert torch.equal(out2.cpu(), (got + got).to(dtype))
    ops.hifi_resblock1(x.to(DEV), wk, bd, (1, 3, 5), out2, K, mode=2, scale=1.0 / 3.0, final_slope=0.01)
    want2 = ((got + got).to(dtype).float() + got) * (1.0 / 3.0)
    assert torch.equal(out2.cpu(), torch.where(want2 > 0, want2, want2 * 0.01).to(dtype))


def test_fused_and_unfused_generators_agree(cfg):
    gen = build(cfg, 5)
    mel = make_mel(2, 40, seed=9).to(DEV)
    gen.fused = True
    wf = gen(mel)
    gen.fused = False
    wu = gen(mel)
    r = rel_rms(wf.cpu(), wu.cpu())
    print("fused vs conv-by-conv generator: rel-RMS %.3f%%" % (100 * r))
    assert r <= 0.002


@pytest.mark.parametrize("K,dil,B,ln,res", [(3, 1, 2, 700, False), (7, 3, 1, 256, True), (11, 5, 2, 1000, True), (11, 1, 1, 5, False), (7, 5, 3, 257, True)])
def test_window_conv_vs_fp64(K, dil, B, ln, res):
    """The C = 128 window-conv kernel against fp64 conv1d on the same fp16-rounded operands (K-scaled summation bound),
    with the residual and the second (activated) output."""
    from tts_king_amd import ops
    import torch.nn.functional as F
    C = 128
    g = torch.Generator().manual_seed(K * 100 + ln)
    x = torch.randn(B, ln, C, generator=g).half()
    w = (torch.randn(C, C, K, generator=g) * (C * K) ** -0.5).half()
    b = 0.1 * torch.randn(C, generator=g)
    r = torch.randn(B, ln, C, generator=g).half() if res else None
    ref = F.conv1d(x.double().transpose(1, 2), w.double(), b.double(), dilation=dil, padding=dil * (K - 1) // 2).transpose(1, 2)
    if res:
        ref = ref + r.double()
    pack = ops.pack_resblock_weight(w.float().to(DEV), dtype=torch.float16)
    out2 = torch.empty(B, ln, C, dtype=torch.float16, device=DEV) if res else None
    out = ops.hifi_conv_window(x.to(DEV), pack, b.to(DEV), K, dil, R=r.to(DEV) if res else None, out2=out2, lrelu_out=not res)
    want = ref if res else torch.where(ref > 0, ref, 0.1 * ref)
    err = float((out.float().cpu().double() - want).abs().max())
    tol = 2e-6 * (C * K) ** 0.5 * float(want.abs().max() + 1) + float(want.abs().max()) * 2 ** -10
    assert err <= tol, (err, tol)
    if res:
        o = out.float().cpu()
        assert torch.equal(out2.cpu(), torch.where(o > 0, o, 0.1 * o).half())


@pytest.mark.parametrize("C,K,dil,B,ln,dt", [(128, 3, 1, 2, 700, torch.float16), (128, 7, 3, 1, 96, torch.float16), (128, 11, 5, 2, 1000, torch.float16),
                                             (128, 11, 1, 1, 5, torch.float16), (128, 7, 5, 3, 257, torch.bfloat16), (128, 3, 5, 1, 193, torch.float16),
                                             (64, 11, 5, 2, 1000, torch.float16), (64, 3, 1, 1, 176, torch.float16), (64, 7, 3, 2, 353, torch.bfloat16),
                                             (64, 11, 3, 1, 7, torch.float16), (32, 11, 5, 2, 900, torch.float16), (32, 7, 1, 1, 177, torch.float16),
                                             (32, 3, 3, 3, 40, torch.bfloat16), (256, 11, 5, 2, 500, torch.float16), (256, 3, 1, 1, 96, torch.float16),
                                             (256, 7, 3, 2, 201, torch.bfloat16), (256, 11, 3, 1, 9, torch.float16)])
def test_conv_pair_equals_two_window_convs(C, K, dil, B, ln, dt):
    """ttsk_hifi_conv_pair (c1 dilated -> lrelu -> c2 -> + x in one launch, lrelu(c1) kept in LDS; hifi/models.py:88-95) is
    bit-identical to the two window-conv launches it replaces (same fp16 roundings, same accumulation order), at tile-multiple,
    ragged and shorter-than-a-tile lengths; and close to fp64 math on the same operands (all three channel counts)."""
    from tts_king_amd import ops
    g = torch.Generator().manual_seed(K * 1000 + ln + C)
    x = torch.randn(B, ln, C, generator=g).to(dt).to(DEV)
    w1 = torch.randn(C, C, K, generator=g) * (C * K) ** -0.5
    w2 = torch.randn(C, C, K, generator=g) * (C * K) ** -0.5
    b1, b2 = (0.1 * torch.randn(C, generator=g)).to(DEV), (0.1 * torch.randn(C, generator=g)).to(DEV)
    p1, p2 = ops.pack_resblock_weight(w1.to(DEV), dtype=dt), ops.pack_resblock_weight(w2.to(DEV), dtype=dt)
    assert ops.hifi_conv_pair_supported(C, K, dil) and not ops.hifi_conv_pair_supported(512, K, dil) and not ops.hifi_conv_pair_supported(C, 13, 1)
    got = ops.hifi_conv_pair(x, p1, b1, p2, b2, K, dil)
    xl = torch.where(x.float() > 0, x.float(), 0.1 * x.float()).to(dt)
    if C == 128:                      # the two launches the pair replaces exist at C = 128 only (C = 64 / 32: the frame-split kernel)
        tl = ops.hifi_conv_window(xl, p1, b1, K, dil, lrelu_out=True)
        want = ops.hifi_conv_window(tl, p2, b2, K, 1, R=x)
        assert torch.equal(got, want), float((got.float() - want.float()).abs().max())
    # fp64 on the same 16-bit operands (t rounded to 16 bits as the kernel does)
    xd = xl.double().cpu().transpose(1, 2)
    t = F.conv1d(xd, w1.to(dt).double(), b1.double().cpu(), dilation=dil, padding=dil * (K - 1) // 2)
    t = torch.where(t > 0, t, 0.1 * t).to(dt).double()
    ref = (F.conv1d(t, w2.to(dt).double(), b2.double().cpu(), padding=(K - 1) // 2) + x.double().cpu().transpose(1, 2)).transpose(1, 2)
    err = float((got.double().cpu() - ref).abs().max())
    eps = 2.0 ** (-10 if dt == torch.float16 else -7)
    assert err <= 4 * eps * float(ref.abs().max()), (err, float(ref.abs().max()))
    # the MRF average folded into the epilogue (ttsk_hifi_resblock1's modes): out += y, out = lrelu((out + y) * scale).  The fp32 y
    # is not observable, so the check recomputes from the fp64 reference within two ulps of the 16-bit type.
    acc = got.clone()
    ops.hifi_conv_pair(x, p1, b1, p2, b2, K, dil, out=acc, mode=1)
    assert float((acc.double().cpu() - (got.double().cpu() + ref)).abs().max()) <= 4 * eps * float(ref.abs().max())
    prev = acc.clone()
    ops.hifi_conv_pair(x, p1, b1, p2, b2, K, dil, out=acc, mode=2, scale=1.0 / 3.0, final_slope=0.01)
    want3 = (prev.double().cpu() + ref) / 3
    want3 = torch.where(want3 > 0, want3, 0.01 * want3)
    assert float((acc.double().cpu() - want3).abs().max()) <= 2 * eps * float(ref.abs().max())


def test_conv_pair_and_window_generators_agree(cfg):
    gen = build(cfg, 5)
    mel = make_mel(2, 40, seed=9).to(DEV)
    gen.conv_pair = gen.conv_pair_small = True
    a = gen(mel)
    gen.conv_pair = gen.conv_pair_small = False        # C = 128: two window launches per pair; C = 64 / 32: the six-conv fused kernel
    b = gen(mel)
    # not bit-equal: the two-launch path takes lrelu(x) of the stage input from the upsampler's fp32 epilogue value, the pair kernel
    # from x as stored (16-bit); one ulp of the 16-bit type on some elements
    r = rel_rms(a.cpu(), b.cpu())
    print("pair vs two-launch window generator: rel-RMS %.4f%%" % (100 * r))
    assert r <= 0.002


def test_window_and_gemm_generators_agree(cfg):
    gen = build(cfg, 5)
    mel = make_mel(2, 40, seed=9).to(DEV)
    gen.window_conv = True
    a = gen(mel)
    gen.window_conv = False
    b = gen(mel)
    r = rel_rms(a.cpu(), b.cpu())
    print("window-conv vs implicit-GEMM generator: rel-RMS %.3f%%" % (100 * r))
    assert r <= 0.002


@pytest.mark.parametrize("Cin,B,T", [(128, 2, 300), (128, 1, 1), (64, 2, 256), (64, 3, 777), (128, 1, 513)])
def test_stream_upsample_vs_fp64(Cin, B, T):
    """ttsk_hifi_upsample2 (stride 2, kernel 4, padding 1; both phases from one read of the input) vs fp64
    ConvTranspose1d on the same fp16 inputs, and bit-equal channel-by-channel structure at the sequence ends (t = 0, T-1)."""
    from tts_king_amd import ops
    Cout = Cin // 2
    g = torch.Generator().manual_seed(Cin + T)
    x = torch.randn(B, T, Cin, generator=g).half()
    w = (torch.randn(Cin, Cout, 4, generator=g) * (2 * Cin) ** -0.5).half()          # torch ConvTranspose1d layout
    bias = torch.randn(Cout, generator=g)
    ref = F.conv_transpose1d(x.double().transpose(1, 2), w.double(), bias.double(), stride=2, padding=1).transpose(1, 2)
    assert ops.hifi_upsample2_supported(Cin, Cout, 2, 4) and not ops.hifi_upsample2_supported(Cin, Cout, 8, 16)
    wp = ops.pack_conv_weight(w.float().to(DEV), transposed=True, dtype=torch.float16)           # (4, Cout, Cin)
    out = ops.hifi_upsample2(x.to(DEV), wp, bias.to(DEV))
    assert out.shape == (B, 2 * T, Cout) and out.dtype == torch.float16
    r = rel_rms(out.float().cpu(), ref.float())
    print("upsample2 Cin=%d B=%d T=%d: rel-RMS %.4f%%" % (Cin, B, T, 100 * r))
    assert r <= 1e-3
    gen = ops.conv_transpose1d(x.to(DEV), wp, bias.to(DEV), 2, 4)                                 # polyphase implicit GEMMs
    assert float((gen.float() - out.float()).abs().max()) <= 2e-3 * float(ref.abs().max())


def test_stream_upsample_generator_agrees(cfg):
    from tts_king_amd.hifi_bench import build_generator
    gen = build_generator(cfg, DEV)
    mel = make_mel(2, 40, seed=3).to(DEV)
    gen.stream_upsample = True
    a = gen(mel)
    gen.stream_upsample = False
    b = gen(mel)
    r = rel_rms(a.cpu(), b.cpu())
    print("streamed vs polyphase-GEMM upsamplers: rel-RMS %.3f%%" % (100 * r))
    assert r <= 2e-3


@pytest.mark.parametrize("Cin,Cout,B,T", [(512, 256, 2, 37), (512, 256, 1, 1), (256, 128, 3, 130), (256, 128, 2, 384), (512, 256, 8, 384), (256, 64, 1, 113)])
def test_window_upsample8_vs_fp64(Cin, Cout, B, T):
    """ttsk_hifi_upsample8 (stride 8, kernel 16, padding 4 as a two-tap window conv with 8 * Cout phase-major channels: hifi/models.py:166-176
    with upsample_rates[i] = 8) vs fp64 ConvTranspose1d on the same fp16 inputs — including the sequence ends, where the taps that reach
    x[-1] / x[T] must see zeros — and vs the polyphase implicit GEMMs it replaces."""
    from tts_king_amd import ops
    g = torch.Generator().manual_seed(Cin + Cout + T)
    x = torch.randn(B, T, Cin, generator=g).half()
    w = (torch.randn(Cin, Cout, 16, generator=g) * (2 * Cin) ** -0.5).half()         # torch ConvTranspose1d layout
    bias = torch.randn(Cout, generator=g)
    ref = F.conv_transpose1d(x.double().transpose(1, 2), w.double(), bias.double(), stride=8, padding=4).transpose(1, 2)
    assert ops.hifi_upsample8_supported(Cin, Cout, 8, 16) and not ops.hifi_upsample8_supported(Cin, Cout, 2, 4)
    wp = ops.pack_conv_weight(w.float().to(DEV), transposed=True, dtype=torch.float16)           # (16, Cout, Cin)
    pack, b8 = ops.hifi_upsample8_pack(wp, bias.to(DEV))
    out = ops.hifi_upsample8(x.to(DEV), pack, b8, Cout)
    assert out.shape == (B, 8 * T, Cout) and out.dtype == torch.float16
    r = rel_rms(out.float().cpu(), ref.float())
    worst = float((out.float().cpu() - ref.float()).abs().max()) / float(ref.abs().max())
    print("upsample8 %d->%d B=%d T=%d: rel-RMS %.4f%%, max %.2e of max |y|" % (Cin, Cout, B, T, 100 * r, worst))
    assert r <= 1e-3 and worst <= 2e-3
    gen = ops.conv_transpose1d(x.to(DEV), wp, bias.to(DEV), 8, 16)                                # polyphase implicit GEMMs
    assert float((gen.float() - out.float()).abs().max()) <= 2e-3 * float(ref.abs().max())


@pytest.mark.gpu
@pytest.mark.parametrize("B,T,k", [(8, 384, 7), (1, 1, 7), (3, 65, 7), (2, 130, 3)])
def test_conv_pre_on_the_window_kernel_vs_fp64(B, T, k):
    """ttsk_hifi_conv_pre_win (hifi/models.py:152,186 + the LeakyReLU of :188): Conv1d(80 -> 512, k) with zero padding at both ends of every
    utterance, the contraction zero-padded to 96 channels, LeakyReLU(0.1) on the way out — vs float64, and vs the implicit GEMM it replaces."""
    from tts_king_amd import ops
    g = torch.Generator().manual_seed(T + k)
    Cin, Cout = 80, 512
    x = torch.randn(B, T, Cin, generator=g).half()
    w = (torch.randn(Cout, Cin, k, generator=g) * (k * Cin) ** -0.5)
    bias = torch.randn(Cout, generator=g)
    w16 = ops.pack_conv_weight(w.to(DEV), dtype=torch.float16)                                  # (Cout, k, Cin)
    ref = F.leaky_relu(F.conv1d(x.double().transpose(1, 2), w16.double().cpu().permute(0, 2, 1), bias.double(), padding=k // 2), 0.1).transpose(1, 2)
    assert ops.hifi_conv_pre_win_supported(Cin, Cout, k) and not ops.hifi_conv_pre_win_supported(Cin, Cout, 11)
    pack = ops.hifi_conv_pre_win_pack(w16)
    out = ops.hifi_conv_pre_win(x.to(DEV), pack, bias.to(DEV), Cout, k, 0.1)
    assert out.shape == (B, T, Cout) and out.dtype == torch.float16
    r = rel_rms(out.float().cpu(), ref.float())
    worst = float((out.float().cpu() - ref.float()).abs().max()) / float(ref.abs().max())
    print("conv_pre window B=%d T=%d k=%d: rel-RMS %.4f%%, max %.2e of max |y|" % (B, T, k, 100 * r, worst))
    assert r <= 1e-3 and worst <= 2e-3
    gen = ops.conv1d(x.to(DEV), w16, bias.to(DEV), flags=ops.LRELU_OUT, out_slope=0.1)
    assert float((gen.float() - out.float()).abs().max()) <= 2e-3 * float(ref.abs().max())


@pytest.mark.gpu
@pytest.mark.parametrize("Cout,B,T", [(128, 2, 300), (128, 1, 1), (128, 3, 96), (128, 2, 97), (128, 8, 3072), (32, 2, 200), (256, 1, 130)])
def test_loop_upsample8_vs_fp64_and_window_kernel(Cout, B, T):
    """ttsk_hifi_upsample_loop (Cin = 256, stride 8: a 96-frame window loaded once, the 8 * Cout / 256 channel groups looped inside the
    workgroup) vs float64 ConvTranspose1d (hifi/models.py:166-176) — ragged last tiles, T = 1, tile-sized and bench-sized inputs, one to
    eight channel groups — and vs ttsk_hifi_upsample_win on the same pack: the same products in the same order with the bias added first
    instead of last, so the fp16 results may differ in the last place only."""
    from tts_king_amd import ops
    Cin = 256
    g = torch.Generator().manual_seed(Cout + T)
    x = torch.randn(B, T, Cin, generator=g).half()
    w = (torch.randn(Cin, Cout, 16, generator=g) * (2 * Cin) ** -0.5).half()
    bias = torch.randn(Cout, generator=g)
    ref = F.conv_transpose1d(x.double().transpose(1, 2), w.double(), bias.double(), stride=8, padding=4).transpose(1, 2)
    assert ops.hifi_upsample_loop_supported(Cin, Cout, 8, 16) and not ops.hifi_upsample_loop_supported(512, Cout, 8, 16)
    assert not ops.hifi_upsample_loop_supported(Cin, Cout, 2, 4)
    wp = ops.pack_conv_weight(w.float().to(DEV), transposed=True, dtype=torch.float16)
    pack, b8 = ops.hifi_upsample_win_pack(wp, bias.to(DEV), 8)
    out = ops.hifi_upsample_loop(x.to(DEV), pack, b8, Cout, 8)
    assert out.shape == (B, 8 * T, Cout) and out.dtype == torch.float16
    r = rel_rms(out.float().cpu(), ref.float())
    worst = float((out.float().cpu() - ref.float()).abs().max()) / float(ref.abs().max())
    print("upsample_loop 256->%d B=%d T=%d: rel-RMS %.4f%%, max %.2e of max |y|" % (Cout, B, T, 100 * r, worst))
    assert r <= 1e-3 and worst <= 2e-3
    if ops.hifi_upsample_win_supported(Cin, Cout, 8, 16):
        win = ops.hifi_upsample_win(x.to(DEV), pack, b8, Cout, 8)
        d = (win.float() - out.float()).abs()
        assert float(d.max()) <= 2.0 ** -10 * float(ref.abs().max()) and float((d > 0).float().mean()) <= 0.05      # an fp16 ulp, on few elements


@pytest.mark.gpu
def test_loop_upsample_generator_agrees(cfg):
    from tts_king_amd.hifi_bench import build_generator
    gen = build_generator(cfg, DEV)
    mel = make_mel(2, 40, seed=5).to(DEV)
    assert gen.loop_upsample
    a = gen(mel)
    gen.loop_upsample = False
    b = gen(mel)
    r = rel_rms(a.cpu(), b.cpu())
    print("looped vs one-group-per-workgroup 256 -> 128 upsampler, waveform: rel-RMS %.4f%%" % (100 * r))
    assert r <= 1e-3


def test_window_upsample8_generator_agrees(cfg):
    from tts_king_amd.hifi_bench import build_generator
    gen = build_generator(cfg, DEV)
    mel = make_mel(2, 40, seed=4).to(DEV)
    gen.window_upsample = True
    gen._packed = None
    a = gen(mel)
    assert any(p is not None for p in gen._packed["ups8"])
    gen.window_upsample = False
    gen._packed = None
    b = gen(mel)
    assert all(p is None for p in gen._packed["ups8"])
    r = rel_rms(a.cpu(), b.cpu())
    print("window-conv vs polyphase-GEMM stride-8 upsamplers: rel-RMS %.3f%%" % (100 * r))
    assert r <= 2e-3


@pytest.mark.parametrize("B,T", [(2, 300), (1, 1), (3, 777), (8, 3000)])
def test_window_upsample2_vs_fp64_and_stream_kernel(B, T):
    """ttsk_hifi_upsample_win at stride 2 (ConvTranspose1d 128 -> 64, kernel 4, padding 1: out[2t] = x[t] W[1] + x[t-1] W[3], out[2t+1] =
    x[t] W[2] + x[t+1] W[0]) vs fp64 on the same fp16 inputs and vs ttsk_hifi_upsample2."""
    from tts_king_amd import ops
    Cin, Cout = 128, 64
    g = torch.Generator().manual_seed(T)
    x = torch.randn(B, T, Cin, generator=g).half()
    w = (torch.randn(Cin, Cout, 4, generator=g) * (2 * Cin) ** -0.5).half()
    bias = torch.randn(Cout, generator=g)
    ref = F.conv_transpose1d(x.double().transpose(1, 2), w.double(), bias.double(), stride=2, padding=1).transpose(1, 2)
    assert ops.hifi_upsample_win_supported(Cin, Cout, 2, 4) and not ops.hifi_upsample_win_supported(64, 32, 2, 4)
    wp = ops.pack_conv_weight(w.float().to(DEV), transposed=True, dtype=torch.float16)
    pack, brep = ops.hifi_upsample_win_pack(wp, bias.to(DEV), 2)
    out = ops.hifi_upsample_win(x.to(DEV), pack, brep, Cout, 2)
    assert out.shape == (B, 2 * T, Cout) and out.dtype == torch.float16
    r = rel_rms(out.float().cpu(), ref.float())
    worst = float((out.float().cpu() - ref.float()).abs().max()) / float(ref.abs().max())
    print("window upsample2 B=%d T=%d: rel-RMS %.4f%%, max %.2e of max |y|" % (B, T, 100 * r, worst))
    assert r <= 1e-3 and worst <= 2e-3
    other = ops.hifi_upsample2(x.to(DEV), wp, bias.to(DEV))
    assert float((other.float() - out.float()).abs().max()) <= 2e-3 * float(ref.abs().max())


def _mrf32_inputs(dtype, B, ln, seed):
    """Seeded stage input and the 18 conv packs / biases + conv_post of a C = 32 last stage (kernel sizes 3, 7, 11)."""
    from tts_king_amd import ops
    g = torch.Generator().manual_seed(seed)
    x = (torch.randn(B, ln, 32, generator=g) * 0.5).to(dtype).to(DEV)
    ks, dil = (3, 7, 11), ((1, 3, 5),) * 3
    ws, bs = [], []
    for k in ks:
        for _ in range(6):
            w = torch.randn(32, 32, k, generator=g) / (32 * k) ** 0.5
            ws.append(ops.pack_resblock_weight(w.to(DEV), dtype=dtype))
            bs.append((0.02 * torch.randn(32, generator=g)).to(DEV))
    wpost = ops.pack_conv_weight((torch.randn(1, 32, 7, generator=g) / 15.0).to(DEV), dtype=dtype)
    bpost = (0.02 * torch.randn(1, generator=g)).to(DEV)
    return x, ks, dil, ws, bs, wpost, bpost


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("B,ln", [(2, 1000), (1, 386), (1, 387), (3, 5), (1, 1), (2, 4099)])
def test_fused_last_stage_equals_four_launches(B, ln, dtype):
    """ttsk_hifi_mrf32_post (csrc/mrf32.hip: three ResBlock1s + average + LeakyReLU(0.01) + conv_post + tanh in one launch;
    hifi/models.py:190-199) against the launches it replaces — ttsk_hifi_resblock1 x 3 in modes 0 / 1 / 2 and ttsk_hifi_conv_post —
    on the same input and packs: the activated average and the waveform BIT-identical (same MFMA order per conv, same 16-bit roundings
    of every block output and partial sum, same conv_post summation order).  Lengths around the 386-frame tile, shorter than the halo,
    and one frame."""
    from tts_king_amd import ops
    x, ks, dil, ws, bs, wpost, bpost = _mrf32_inputs(dtype, B, ln, 5 + ln)
    ref = torch.empty_like(x)
    for j, k in enumerate(ks):
        ops.hifi_resblock1(x, ws[6 * j:6 * j + 6], bs[6 * j:6 * j + 6], dil[j], ref, k, mode=0 if j == 0 else (2 if j == 2 else 1),
                           scale=1.0 / 3.0, slope=0.1, final_slope=0.01 if j == 2 else 1.0)
    want = ops.hifi_conv_post(ref, wpost, bpost)
    stage = torch.full_like(x, 7.0)
    got = ops.hifi_mrf32_post(x, ws, bs, dil, ks, wpost, bpost, stage_out=stage)
    torch.cuda.synchronize()
    assert got.shape == want.shape == (B, 1, ln)
    assert torch.equal(stage.view(torch.int16), ref.view(torch.int16)), "activated MRF average differs: %d of %d elements" % (
        int((stage.view(torch.int16) != ref.view(torch.int16)).sum()), stage.numel())
    assert torch.equal(got, want), "waveform differs: max abs %.3e" % float((got - want).abs().max())


def test_fused_last_stage_full_size_and_generator(cfg):
    """The same at the BASELINE.json configs[2] size (8 x 98,304 frames), and the generator with and without the fused last stage:
    identical waveforms."""
    from tts_king_amd import ops
    x, ks, dil, ws, bs, wpost, bpost = _mrf32_inputs(torch.float16, 8, 98304, 77)
    ref = torch.empty_like(x)
    for j, k in enumerate(ks):
        ops.hifi_resblock1(x, ws[6 * j:6 * j + 6], bs[6 * j:6 * j + 6], dil[j], ref, k, mode=0 if j == 0 else (2 if j == 2 else 1),
                           scale=1.0 / 3.0, slope=0.1, final_slope=0.01 if j == 2 else 1.0)
    want = ops.hifi_conv_post(ref, wpost, bpost)
    got = ops.hifi_mrf32_post(x, ws, bs, dil, ks, wpost, bpost)
    torch.cuda.synchronize()
    assert torch.equal(got, want)
    gen = build(cfg, 11)
    mel = make_mel(2, 50, seed=9).to(DEV)
    assert gen.mrf_fused
    a = gen(mel)
    gen.mrf_fused = False
    b = gen(mel)
    assert torch.equal(a, b)


WS_PAIR_CASES = [(3, 1, 2, 700, torch.float16), (3, 3, 1, 192, torch.float16), (3, 5, 3, 193, torch.bfloat16), (7, 1, 1, 191, torch.float16),
                 (7, 3, 2, 1000, torch.float16), (7, 5, 2, 385, torch.bfloat16), (11, 1, 1, 5, torch.float16), (11, 3, 3, 383, torch.float16),
                 (11, 5, 2, 1000, torch.float16), (11, 5, 1, 1, torch.bfloat16), (7, 3, 8, 4099, torch.float16), (11, 5, 4, 12288, torch.float16),
                 (3, 1, 2, 700, torch.float16, 128), (3, 3, 1, 96, torch.float16, 128), (3, 5, 3, 97, torch.bfloat16, 128),
                 (3, 1, 1, 3, torch.bfloat16, 128), (3, 3, 3, 191, torch.float16, 128), (3, 5, 8, 4099, torch.float16, 128)]


@pytest.mark.parametrize("K,dil,B,ln,dt,C", [c if len(c) == 6 else c + (64,) for c in WS_PAIR_CASES])
def test_weights_stationary_pair_is_bit_identical_to_the_pair_kernel(K, dil, B, ln, dt, C):
    """ttsk_hifi_conv_pair_ws (round 6, csrc/pairws.hip: persistent workgroups, both convs' weights in registers, c1 of tile s beside c2 of
    tile s - 1; hifi/models.py:88-95) against ttsk_hifi_conv_pair at C = 64 (every kernel size) and C = 128 (k = 3): same roundings and accumulation order, so bit-identical — in all three
    MRF modes (:190-197), at one tile, ragged tiles, more tiles than workgroups (every workgroup walks several, runs crossing utterances) and with the
    grid capped to 1, 3 and 7 workgroups (uneven runs); and against fp64 on the same 16-bit operands."""
    from tts_king_amd import ops
    g = torch.Generator().manual_seed(K * 1000 + ln + dil)
    x = torch.randn(B, ln, C, generator=g).to(dt).to(DEV)
    w1 = torch.randn(C, C, K, generator=g) * (C * K) ** -0.5
    w2 = torch.randn(C, C, K, generator=g) * (C * K) ** -0.5
    b1, b2 = (0.1 * torch.randn(C, generator=g)).to(DEV), (0.1 * torch.randn(C, generator=g)).to(DEV)
    p1, p2 = ops.pack_resblock_weight(w1.to(DEV), dtype=dt), ops.pack_resblock_weight(w2.to(DEV), dtype=dt)
    assert ops.hifi_conv_pair_ws_supported(C, K, dil, ln) and not ops.hifi_conv_pair_ws_supported(128, 7, dil) and not ops.hifi_conv_pair_ws_supported(256, K, dil)
    assert not ops.hifi_conv_pair_ws_supported(C, 5, 1)
    want = ops.hifi_conv_pair(x, p1, b1, p2, b2, K, dil)
    for cap in (0, 1, 3, 7):
        got = ops.hifi_conv_pair(x, p1, b1, p2, b2, K, dil, ws=True, max_wgs=cap)
        assert torch.equal(got, want), (cap, float((got.float() - want.float()).abs().max()))
    prev = torch.randn(B, ln, C, generator=g).to(dt).to(DEV)
    for mode, kw in ((1, {}), (2, dict(scale=1.0 / 3.0, final_slope=0.01))):
        a, b = prev.clone(), prev.clone()
        ops.hifi_conv_pair(x, p1, b1, p2, b2, K, dil, out=a, mode=mode, **kw)
        ops.hifi_conv_pair(x, p1, b1, p2, b2, K, dil, out=b, mode=mode, ws=True, **kw)
        assert torch.equal(a, b), (mode, float((a.float() - b.float()).abs().max()))
    if B * ln <= 4000:
        xl = torch.where(x.float() > 0, x.float(), 0.1 * x.float()).to(dt)
        xd = xl.double().cpu().transpose(1, 2)
        t = F.conv1d(xd, w1.to(dt).double(), b1.double().cpu(), dilation=dil, padding=dil * (K - 1) // 2)
        t = torch.where(t > 0, t, 0.1 * t).to(dt).double()
        ref = (F.conv1d(t, w2.to(dt).double(), b2.double().cpu(), padding=(K - 1) // 2) + x.double().cpu().transpose(1, 2)).transpose(1, 2)
        eps = 2.0 ** (-10 if dt == torch.float16 else -7)
        assert float((want.double().cpu() - ref).abs().max()) <= 4 * eps * float(ref.abs().max())


def test_generator_with_and_without_the_weights_stationary_pairs_agrees_bit_for_bit(cfg):
    """The whole generator with the C = 64 stage's k = 7 / 11 pairs on either kernel: bit-identical waveform (k = 3 moves from the six-conv fused kernel
    to three pair launches, which round the block's intermediate tensors to 16 bits: that stage is compared within the fused-vs-pairs bar)."""
    gen = build(cfg, 5)
    mel = make_mel(2, 40, seed=9).to(DEV)
    gen.pair_ws, gen.pair_ws_min_tiles = True, 1          # (the test's stage tensors are far below the size from which the kernel is the default)
    a = gen(mel)
    gen.pair_ws = False
    b = gen(mel)
    torch.cuda.synchronize()
    assert rel_rms(a.float().cpu(), b.float().cpu()) <= 2e-3
