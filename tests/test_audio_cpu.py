"""CPU: the mel-extraction oracle (oracle/audio.py) against its pins — librosa's published known answers for
the Slaney mel scale / filterbank, the committed torch.stft golden, and an independent float64 numpy rfft framing — and
the host-side filterbank of the product (tts_king_amd/audio.py) against the oracle's."""
import os

import numpy as np
import torch

from oracle import audio as OA

GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "mel_extraction.npz"))
N_FFT, HOP, WIN, N_MEL, SR, FMIN, FMAX = (int(v) for v in GOLD["params"])

# librosa.mel_frequencies(n_mels=40) as printed in its docstring (librosa 0.7.2, core/time_frequency.py)
LIBROSA_MEL_FREQS_40 = [0., 85.317, 170.635, 255.952, 341.269, 426.586, 511.904, 597.221, 682.538, 767.855, 853.173, 938.49,
                        1024.856, 1119.114, 1222.042, 1334.436, 1457.167, 1591.187, 1737.532, 1897.337, 2071.84, 2262.393,
                        2470.47, 2697.686, 2945.799, 3216.731, 3512.582, 3835.643, 4188.417, 4573.636, 4994.285, 5453.621,
                        5955.205, 6502.92, 7101.009, 7754.107, 8467.272, 9246.028, 10096.408, 11025.]


def test_mel_frequencies_known_answer():
    got = OA.mel_frequencies(40, 0.0, 11025.0)
    assert np.allclose(got, LIBROSA_MEL_FREQS_40, atol=6e-4)


def test_filterbank_known_answer_and_shape_properties():
    w = OA.mel_filterbank(22050, 2048, 128, 0.0, None)
    assert w.shape == (128, 1025) and w.dtype == np.float32
    assert round(float(w[0][1]), 3) == 0.016 and w[0][0] == 0.0          # librosa.filters.mel docstring: [0., 0.016, ...
    w = OA.mel_filterbank(SR, N_FFT, N_MEL, FMIN, FMAX).astype(np.float64)
    assert (w >= 0).all() and (w.sum(1) > 0).all()
    # Slaney normalisation: every triangle has unit area in Hz (sampled at the FFT bin spacing)
    area = w.sum(1) * (SR / N_FFT)
    assert np.allclose(area[5:], 1.0, atol=0.12)
    # filters above fmax are empty
    assert w[:, int(FMAX / (SR / N_FFT)) + 2:].sum() == 0.0


def test_oracle_matches_golden():
    y = torch.from_numpy(GOLD["y"])
    mel = OA.mel_spectrogram(y, N_FFT, N_MEL, SR, HOP, WIN, FMIN, FMAX)
    assert mel.shape == (2, N_MEL, y.shape[1] // HOP)
    assert np.allclose(mel.numpy(), GOLD["mel_hifi"], atol=1e-5)
    mel_t, energy = OA.tacotron_mel(y, N_FFT, HOP, WIN, N_MEL, SR, FMIN, FMAX)
    assert mel_t.shape == (2, N_MEL, 1 + y.shape[1] // HOP)
    assert np.allclose(mel_t.numpy(), GOLD["mel_taco"], atol=1e-5) and np.allclose(energy.numpy(), GOLD["energy"], rtol=1e-5)


def test_oracle_against_numpy_rfft():
    """Independent float64 restatement: explicit framing + numpy rfft."""
    y = GOLD["y"].astype(np.float64)
    win = torch.hann_window(WIN, dtype=torch.float64).numpy()
    fb = OA.mel_filterbank(SR, N_FFT, N_MEL, FMIN, FMAX).astype(np.float64)
    for pad, eps, key in (((N_FFT - HOP) // 2, 1e-9, "mel_hifi"), (N_FFT // 2, 0.0, "mel_taco")):
        for b in range(y.shape[0]):
            yp = np.pad(y[b], (pad, pad), mode="reflect")
            T = (len(yp) - N_FFT) // HOP + 1
            fr = np.stack([yp[t * HOP: t * HOP + N_FFT] * win for t in range(T)])
            mag2 = np.abs(np.fft.rfft(fr, axis=1)) ** 2
            ref = np.log(np.maximum(fb @ np.sqrt(mag2 + eps).T, 1e-5))
            assert np.abs(ref - GOLD[key][b]).max() < 2e-4
            if key == "mel_taco":
                assert np.allclose(np.sqrt(mag2.sum(1)), GOLD["energy"][b], rtol=1e-4)


def test_product_filterbank_equals_oracle():
    from tts_king_amd.audio import slaney_mel_filterbank
    for args in ((SR, N_FFT, N_MEL, FMIN, FMAX), (22050, 2048, 128, 0.0, None), (16000, 512, 40, 50.0, 7600.0)):
        assert np.array_equal(slaney_mel_filterbank(*args), OA.mel_filterbank(*args)) or \
            np.allclose(slaney_mel_filterbank(*args), OA.mel_filterbank(*args), atol=1e-8)
