"""GPU: HiFi-GAN generator on the HIP kernels against (a) the waveform the reference produced (tests/golden/
hifi_b2_t32.npz) and (b) the oracle on the same inputs, through the reference's surface (weight-normed
state_dict -> remove_weight_norm -> forward; HIFIapi.generate -> int16).

Stated tolerance (bf16 activations/weights, fp32 accumulate, vs the reference's fp32; the reference itself under
bf16 autocast measured rel-RMS 0.35 %): waveform rel-RMS <= 1 %, max-abs <= 0.02 of full
scale; int16 samples within 1 % of full scale of the reference's; weight-norm folding fp32-exact (rtol 1e-5);
the int16 cast of an identical float input: bit-exact."""
import copy
import os

import numpy as np
import pytest
import torch

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
            np.testing.assert_allclose(float(sd[name[9:]].double().norm()), float(g[name]), rtol=1e-5)


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
    assert r <= 0.01 and a <= 0.02


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
    assert np.abs(i16.astype(np.int32) - g["int16"].astype(np.int32)).max() <= 328          # 1 % of full scale
    # the cast itself is exact: same float input -> same int16 as numpy astype (truncation toward zero)
    wav = api(mel)
    assert np.array_equal(ohifi.to_int16(wav.cpu(), c.hifi.MAX_WAV_VALUE), api.generate(mel))
    with pytest.raises(NotImplementedError):
        api.train()
    c.model_config["vocoder"]["use_cpu"] = True
    from tts_king_amd.lib import TtskError
    with pytest.raises(TtskError):
        hifiapi.HIFIapi(c, "cuda:0")


@pytest.mark.parametrize("B,T", [(1, 1), (3, 7), (1, 100), (2, 384)])
def test_waveform_vs_oracle_shapes(cfg, B, T):
    """Ragged / tiny / BASELINE-size time lengths against the oracle (same seeded weights)."""
    sdw = hifi_state_dict_wn(11)
    gen = build(cfg, 11)
    mel = make_mel(B, T, seed=100 + T)
    with torch.no_grad():
        want = ohifi.generator(ohifi.fold_weight_norm(sdw), cfg.hifi, mel)
    got = gen(mel.to(DEV)).cpu()
    assert got.shape == want.shape == (B, 1, 256 * T)
    r = rel_rms(got, want)
    print("B=%d T=%d rel-RMS %.3f%% max-abs %.5f" % (B, T, 100 * r, float((got - want).abs().max())))
    assert r <= 0.01 and float((got - want).abs().max()) <= 0.02


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
    assert torch.equal(single, w1[5:6])
