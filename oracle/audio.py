"""ORACLE (test infrastructure, not product code): CPU restatement of the reference's mel-spectrogram extraction
(SURVEY.md §8 row f-3).

Two variants exist in the reference and both are restated:
  * `mel_spectrogram`  — hifi/meldataset.py:49-74: reflect-pad (n_fft - hop)/2, torch.stft(center=False, Hann),
    sqrt(re^2 + im^2 + 1e-9), mel basis matmul, log(clamp(., 1e-5)).
  * `tacotron_mel`     — fs_two/audio/stft.py:57-90 (STFT.transform as a strided conv with a windowed Fourier basis,
    reflect-pad n_fft/2) and :174-193 (TacotronSTFT.mel_spectrogram: mel + log-clamp, energy = L2 norm over frequency).

Third-party algorithm: the mel filterbank is `librosa.filters.mel` (reference pin: librosa == 0.7.2, requirements.txt:3;
called as `librosa_mel_fn(sampling_rate, n_fft, num_mels, fmin, fmax)`, i.e. htk=False, norm=1 = Slaney area
normalisation).  librosa is not in this image, so its published algorithm is restated in `mel_frequencies` /
`mel_filterbank`.

Pinning: neither reference function runs in this container as written (meldataset.py calls torch.stft without the
`return_complex` argument current torch requires; stft.py hard-codes `.cuda(3)`), so this file is pinned by
  (1) the known answers printed in librosa's own docstrings (`mel_frequencies(n_mels=40)`, `filters.mel(22050, 2048)[0][1]`),
  (2) `torch.stft` itself — the third-party op the reference calls — with the reference's arguments
      (tests/golden/mel_extraction.npz, tools/make_goldens.py:g11_mel),
  (3) an independent float64 numpy rfft framing in tests/test_audio_cpu.py.
"""
import numpy as np
import torch
import torch.nn.functional as F


def hz_to_mel(f):
    """librosa.core.time_frequency.hz_to_mel (htk=False): linear below 1 kHz (200/3 Hz per mel), log above."""
    f = np.asarray(f, dtype=np.float64)
    f_sp = 200.0 / 3
    min_log_hz = 1000.0
    min_log_mel = min_log_hz / f_sp
    logstep = np.log(6.4) / 27.0
    mel = f / f_sp
    return np.where(f >= min_log_hz, min_log_mel + np.log(np.maximum(f, 1e-30) / min_log_hz) / logstep, mel)


def mel_to_hz(m):
    """librosa.core.time_frequency.mel_to_hz (htk=False)."""
    m = np.asarray(m, dtype=np.float64)
    f_sp = 200.0 / 3
    min_log_hz = 1000.0
    min_log_mel = min_log_hz / f_sp
    logstep = np.log(6.4) / 27.0
    return np.where(m >= min_log_mel, min_log_hz * np.exp(logstep * (m - min_log_mel)), f_sp * m)


def mel_frequencies(n_mels, fmin, fmax):
    """librosa.mel_frequencies: n_mels points uniformly spaced on the (Slaney) mel axis."""
    return mel_to_hz(np.linspace(hz_to_mel(fmin), hz_to_mel(fmax), n_mels))


def mel_filterbank(sr, n_fft, n_mels, fmin, fmax):
    """librosa.filters.mel(sr, n_fft, n_mels, fmin, fmax, htk=False, norm=1) -> float32 (n_mels, 1 + n_fft // 2):
    triangles between consecutive mel points, each scaled by 2 / (f[i+2] - f[i])."""
    if fmax is None:
        fmax = sr / 2.0
    fftfreqs = np.linspace(0.0, sr / 2.0, 1 + n_fft // 2)
    mel_f = mel_frequencies(n_mels + 2, fmin, fmax)
    fdiff = np.diff(mel_f)
    ramps = mel_f[:, None] - fftfreqs[None, :]
    w = np.zeros((n_mels, 1 + n_fft // 2))
    for i in range(n_mels):
        lower = -ramps[i] / fdiff[i]
        upper = ramps[i + 2] / fdiff[i + 1]
        w[i] = np.maximum(0.0, np.minimum(lower, upper))
    w *= (2.0 / (mel_f[2:n_mels + 2] - mel_f[:n_mels]))[:, None]
    return w.astype(np.float32)


def mel_spectrogram(y, n_fft, num_mels, sampling_rate, hop_size, win_size, fmin, fmax, center=False):
    """hifi/meldataset.py:49-74.  y (B, L) fp32 in [-1, 1] -> (B, num_mels, L // hop) log-mel."""
    basis = torch.from_numpy(mel_filterbank(sampling_rate, n_fft, num_mels, fmin, fmax))
    window = torch.hann_window(win_size)
    p = int((n_fft - hop_size) / 2)
    y = F.pad(y.unsqueeze(1), (p, p), mode="reflect").squeeze(1)
    spec = torch.stft(y, n_fft, hop_length=hop_size, win_length=win_size, window=window, center=center,
                      pad_mode="reflect", normalized=False, onesided=True, return_complex=True)
    spec = torch.sqrt(spec.real.pow(2) + spec.imag.pow(2) + 1e-9)
    spec = torch.matmul(basis, spec)
    return torch.log(torch.clamp(spec, min=1e-5))


def fourier_basis(filter_length, win_length):
    """fs_two/audio/stft.py:25-50: rows = [Re; Im] of the DFT matrix (cutoff = n/2 + 1 each), times the periodic Hann
    window zero-padded to filter_length (scipy get_window('hann', fftbins=True) == torch.hann_window(periodic))."""
    fb = np.fft.fft(np.eye(filter_length))
    cutoff = filter_length // 2 + 1
    fb = np.vstack([np.real(fb[:cutoff]), np.imag(fb[:cutoff])])
    win = torch.hann_window(win_length, periodic=True, dtype=torch.float64).numpy()
    lpad = (filter_length - win_length) // 2
    win = np.pad(win, (lpad, filter_length - win_length - lpad))
    return torch.from_numpy(fb * win[None, :]).float()


def tacotron_mel(y, filter_length, hop_length, win_length, n_mel_channels, sampling_rate, mel_fmin, mel_fmax):
    """fs_two/audio/stft.py:57-90 + :174-193.  y (B, L) -> (log-mel (B, n_mel, 1 + L // hop), energy (B, 1 + L // hop))."""
    basis = fourier_basis(filter_length, win_length)[:, None, :]
    p = filter_length // 2
    x = F.pad(y.unsqueeze(1), (p, p), mode="reflect")
    ft = F.conv1d(x, basis, stride=hop_length, padding=0)
    cutoff = filter_length // 2 + 1
    mag = torch.sqrt(ft[:, :cutoff] ** 2 + ft[:, cutoff:] ** 2)
    mel_basis = torch.from_numpy(mel_filterbank(sampling_rate, filter_length, n_mel_channels, mel_fmin, mel_fmax))
    mel = torch.log(torch.clamp(torch.matmul(mel_basis, mag), min=1e-5))
    energy = torch.norm(mag, dim=1)
    return mel, energy
