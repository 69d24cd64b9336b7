"""Mel-spectrogram extraction on the GPU, behind the reference's two call surfaces:

  * `mel_spectrogram(y, n_fft, num_mels, sampling_rate, hop_size, win_size, fmin, fmax, center=False)`
    — hifi/meldataset.py:49-74 (what the vocoder is trained on and what `hifiapi` consumers feed it);
  * `TacotronSTFT(...).mel_spectrogram(y) -> (mel, energy)` — fs_two/audio/stft.py:145-193 (FS2 preprocessing features).

Pipeline (all on the current stream, no host math after construction): `ttsk_stft_frames` (reflect pad, hop-block rows,
fp16 hi/lo split) -> one `ttsk_gemm` conv launch with the windowed Fourier basis (hi*hi + hi*lo + lo*hi as one contraction,
fp32 accumulate: the reference's own conv-STFT, stft.py:77-84) -> `ttsk_mel_from_spec` (magnitude, Slaney mel filterbank,
log-clamp, energy).  The filterbank is librosa 0.7.2's `filters.mel(htk=False, norm=1)` (reference requirements.txt:3),
built here from its published definition because librosa is not a dependency of this package."""
import numpy as np
import torch

from . import ops

_SIG_SCALE = 4096.0      # powers of two: keep the fp16 low parts of signal and basis in the normal range
_BASIS_SCALE = 1024.0


def _hz_to_mel(f):
    f = np.asarray(f, dtype=np.float64)
    lin = f * 3.0 / 200.0
    return np.where(f >= 1000.0, 15.0 + np.log(np.maximum(f, 1e-30) / 1000.0) * 27.0 / np.log(6.4), lin)


def _mel_to_hz(m):
    m = np.asarray(m, dtype=np.float64)
    return np.where(m >= 15.0, 1000.0 * np.exp(np.log(6.4) / 27.0 * (m - 15.0)), m * 200.0 / 3.0)


def slaney_mel_filterbank(sr, n_fft, n_mels, fmin, fmax):
    """(n_mels, 1 + n_fft//2) float32: triangular filters on the Slaney mel scale, each with unit area in Hz."""
    fmax = sr / 2.0 if fmax is None else fmax
    freqs = np.linspace(0.0, sr / 2.0, 1 + n_fft // 2)
    pts = _mel_to_hz(np.linspace(_hz_to_mel(fmin), _hz_to_mel(fmax), n_mels + 2))
    up = (freqs[None, :] - pts[:-2, None]) / (pts[1:-1] - pts[:-2])[:, None]
    down = (pts[2:, None] - freqs[None, :]) / (pts[2:] - pts[1:-1])[:, None]
    w = np.maximum(0.0, np.minimum(up, down)) * (2.0 / (pts[2:] - pts[:-2]))[:, None]
    return w.astype(np.float32)


class MelExtractor:
    """Device-resident bases + the launch sequence.  `pad` samples are reflected on each side; `eps` is added under
    the magnitude's square root (1e-9 in hifi/meldataset.py:69, 0 in fs_two/audio/stft.py:86)."""

    def __init__(self, n_fft, num_mels, sampling_rate, hop_size, win_size, fmin, fmax, pad, eps, device="cuda:0"):
        if n_fft % hop_size or hop_size % 8 or win_size > n_fft or num_mels > 80 or n_fft > 2048:
            raise ValueError("MelExtractor needs n_fft % hop_size == 0, hop_size % 8 == 0, win_size <= n_fft <= 2048, num_mels <= 80")
        self.n_fft, self.hop, self.num_mels, self.pad, self.eps = n_fft, hop_size, num_mels, int(pad), float(eps)
        self.taps = n_fft // hop_size
        self.nbins = n_fft // 2 + 1
        self.device = torch.device(device)
        # windowed Fourier basis, rows [Re bins | Im bins | zero rows up to a multiple of 8]  (stft.py:25-50)
        n = np.arange(n_fft, dtype=np.float64)
        ang = 2.0 * np.pi * np.outer(np.arange(self.nbins, dtype=np.float64), n) / n_fft
        win = torch.hann_window(win_size, periodic=True, dtype=torch.float64).numpy()
        lpad = (n_fft - win_size) // 2
        win = np.pad(win, (lpad, n_fft - win_size - lpad))
        self.cout = (2 * self.nbins + 7) // 8 * 8
        basis = np.zeros((self.cout, n_fft), dtype=np.float64)
        basis[: self.nbins] = np.cos(ang) * win
        basis[self.nbins: 2 * self.nbins] = -np.sin(ang) * win
        b = torch.from_numpy(basis * _BASIS_SCALE).float().view(self.cout, self.taps, hop_size)
        b_hi = b.half()
        b_lo = (b - b_hi.float()).half()
        # per tap [hi | lo | hi] against the signal rows' [hi | hi | lo]: one contraction = hi*hi + hi*lo + lo*hi
        self.basis = torch.cat([b_hi, b_lo, b_hi], dim=2).contiguous().to(self.device)      # (cout, taps, 3*hop)
        # packed mel filterbank
        fb = slaney_mel_filterbank(sampling_rate, n_fft, num_mels, fmin, fmax)
        start, off, vals = [], [0], []
        for m in range(num_mels):
            nz = np.nonzero(fb[m])[0]
            lo, hi = (int(nz[0]), int(nz[-1]) + 1) if len(nz) else (0, 0)
            start.append(lo)
            vals.append(fb[m, lo:hi])
            off.append(off[-1] + hi - lo)
        self.nnz = off[-1]
        self.fb_vals = torch.from_numpy(np.concatenate(vals) if self.nnz else np.zeros(1, np.float32)).to(self.device)
        self.fb_start = torch.tensor(start, dtype=torch.int32, device=self.device)
        self.fb_off = torch.tensor(off, dtype=torch.int32, device=self.device)

    def frames(self, n_samples):
        return 1 + (n_samples + 2 * self.pad - self.n_fft) // self.hop

    def __call__(self, y):
        """y (B, L) fp32 on the device, |y| <= 1 -> (log-mel (B, num_mels, T), energy (B, T))."""
        if y.dim() != 2 or y.dtype != torch.float32:
            raise ValueError("MelExtractor: y must be (B, L) fp32")
        Bsz, n = y.shape
        T = self.frames(n)
        if T < 1:
            raise ValueError("MelExtractor: signal shorter than one frame")
        rows = T + self.taps - 1
        sig = ops.stft_frames(y.contiguous(), self.pad, rows, self.hop, _SIG_SCALE)
        spec = torch.empty(Bsz * rows, self.cout, dtype=torch.float32, device=y.device)
        K = 3 * self.hop
        ops.gemm(sig, self.basis, spec, Bsz * rows, self.cout, K, K, self.taps * K, self.cout,
                 alpha=1.0 / (_SIG_SCALE * _BASIS_SCALE), taps=self.taps, seg_len=rows, tap_shift0=0, tap_dshift=1, b_tap_stride=K)
        return ops.mel_from_spec(spec, Bsz, rows, T, self.nbins, self.fb_vals, self.fb_start, self.fb_off, self.nnz,
                                 self.num_mels, self.eps)


_extractors = {}


def mel_spectrogram(y, n_fft, num_mels, sampling_rate, hop_size, win_size, fmin, fmax, center=False):
    """Drop-in for hifi/meldataset.py:49-74: y (B, L) fp32 CUDA tensor -> log-mel (B, num_mels, L // hop_size)."""
    if center:
        raise ValueError("mel_spectrogram: the reference only ever calls this with center=False")
    key = (n_fft, num_mels, sampling_rate, hop_size, win_size, fmin, fmax, str(y.device))
    if key not in _extractors:
        _extractors[key] = MelExtractor(n_fft, num_mels, sampling_rate, hop_size, win_size, fmin, fmax,
                                        pad=(n_fft - hop_size) // 2, eps=1e-9, device=y.device)
    return _extractors[key](y)[0]


class TacotronSTFT:
    """Drop-in for fs_two/audio/stft.py:145-193 (constructor arguments and `mel_spectrogram(y) -> (mel, energy)`)."""

    def __init__(self, filter_length, hop_length, win_length, n_mel_channels, sampling_rate, mel_fmin, mel_fmax, device="cuda:0"):
        self.n_mel_channels, self.sampling_rate = n_mel_channels, sampling_rate
        self._ex = MelExtractor(filter_length, n_mel_channels, sampling_rate, hop_length, win_length, mel_fmin, mel_fmax,
                                pad=filter_length // 2, eps=0.0, device=device)

    def mel_spectrogram(self, y):
        if float(y.min()) < -1.0 or float(y.max()) > 1.0:      # stft.py:185-186
            raise AssertionError("TacotronSTFT.mel_spectrogram: input outside [-1, 1]")
        return self._ex(y)
