"""GPU: mel-spectrogram extraction (tts_king_amd/audio.py -> ttsk_stft_frames, ttsk_gemm, ttsk_mel_from_spec)
against the committed golden and the oracle (oracle/audio.py).  Stated tolerance: the HIP path contracts fp16 hi/lo splits
with fp32 accumulation, the reference is fp32 FFT — log-mel within 1e-4 absolute (measured <= 2e-5;
near the 1e-5 clamp log amplifies differences of a few 1e-9 in the linear mel), energy within 1e-4 relative."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "mel_extraction.npz"))
N_FFT, HOP, WIN, N_MEL, SR, FMIN, FMAX = (int(v) for v in GOLD["params"])


def report(tag, got, ref):
    d = np.abs(got - ref)
    print("%s: max-abs %.2e, mean-abs %.2e" % (tag, d.max(), d.mean()))
    return d.max()


def test_hifigan_mel_vs_golden():
    from tts_king_amd.audio import mel_spectrogram
    y = torch.from_numpy(GOLD["y"]).to(DEV)
    mel = mel_spectrogram(y, N_FFT, N_MEL, SR, HOP, WIN, FMIN, FMAX, center=False)
    assert mel.shape == GOLD["mel_hifi"].shape and mel.dtype == torch.float32
    assert report("hifi log-mel", mel.cpu().numpy(), GOLD["mel_hifi"]) < 1e-4


def test_tacotron_mel_and_energy_vs_golden():
    from tts_king_amd.audio import TacotronSTFT
    stft = TacotronSTFT(N_FFT, HOP, WIN, N_MEL, SR, FMIN, FMAX, device=DEV)
    mel, energy = stft.mel_spectrogram(torch.from_numpy(GOLD["y"]).to(DEV))
    assert mel.shape == GOLD["mel_taco"].shape and energy.shape == GOLD["energy"].shape
    assert report("tacotron log-mel", mel.cpu().numpy(), GOLD["mel_taco"]) < 1e-4
    assert np.allclose(energy.cpu().numpy(), GOLD["energy"], rtol=1e-4, atol=1e-5)
    with pytest.raises(AssertionError):
        stft.mel_spectrogram(torch.full((1, 4096), 1.5, device=DEV))


@pytest.mark.parametrize("Bsz,n", [(1, 1024), (3, 5000), (2, 22050), (5, 256 * 423)])
def test_vs_oracle_ragged_sizes(Bsz, n):
    """Lengths that are not multiples of the hop, one-frame inputs, the FS2 batch's 423 frames."""
    from oracle import audio as OA
    from tts_king_amd.audio import mel_spectrogram, TacotronSTFT
    g = torch.Generator().manual_seed(n)
    y = (torch.rand(Bsz, n, generator=g) * 2 - 1) * torch.linspace(1.0, 0.001, n)[None, :]
    ref = OA.mel_spectrogram(y, N_FFT, N_MEL, SR, HOP, WIN, FMIN, FMAX)
    got = mel_spectrogram(y.to(DEV), N_FFT, N_MEL, SR, HOP, WIN, FMIN, FMAX)
    assert got.shape == ref.shape
    assert report("B=%d n=%d hifi" % (Bsz, n), got.cpu().numpy(), ref.numpy()) < 1e-4
    ref_m, ref_e = OA.tacotron_mel(y, N_FFT, HOP, WIN, N_MEL, SR, FMIN, FMAX)
    got_m, got_e = TacotronSTFT(N_FFT, HOP, WIN, N_MEL, SR, FMIN, FMAX, device=DEV).mel_spectrogram(y.to(DEV))
    assert got_m.shape == ref_m.shape
    assert report("B=%d n=%d tacotron" % (Bsz, n), got_m.cpu().numpy(), ref_m.numpy()) < 1e-4
    assert np.allclose(got_e.cpu().numpy(), ref_e.numpy(), rtol=1e-4, atol=1e-5)


def test_silence_and_full_scale():
    """Edge cases: all-zero input sits on the clamp (log 1e-5 for hifi's eps-floor it is log of the eps magnitude's mel);
    a full-scale square wave does not overflow the scaled fp16 operands."""
    from oracle import audio as OA
    from tts_king_amd.audio import mel_spectrogram
    y = torch.zeros(2, 4096)
    y[1] = torch.sign(torch.sin(torch.arange(4096) * 0.05))
    ref = OA.mel_spectrogram(y, N_FFT, N_MEL, SR, HOP, WIN, FMIN, FMAX)
    got = mel_spectrogram(y.to(DEV), N_FFT, N_MEL, SR, HOP, WIN, FMIN, FMAX).cpu()
    assert torch.isfinite(got).all()
    assert report("silence + square", got.numpy(), ref.numpy()) < 1e-4


def test_linearity_property_full_size():
    """Size-independent property at a long input (60 s): the magnitude spectrum is homogeneous, so scaling the input by c
    shifts every un-clamped log-mel value by log c, and the energy by the factor c."""
    from tts_king_amd.audio import TacotronSTFT
    n = 22050 * 60
    g = torch.Generator().manual_seed(3)
    y = ((torch.rand(1, n, generator=g) * 2 - 1) * 0.5).to(DEV)
    stft = TacotronSTFT(N_FFT, HOP, WIN, N_MEL, SR, FMIN, FMAX, device=DEV)
    m1, e1 = stft.mel_spectrogram(y)
    m2, e2 = stft.mel_spectrogram(y * 0.25)
    assert m1.shape == (1, N_MEL, 1 + n // HOP)
    assert float((m1 - m2 - np.log(4.0)).abs().max()) < 1e-3
    assert torch.allclose(e1, e2 * 4.0, rtol=1e-4)
