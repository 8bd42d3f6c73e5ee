"""CPU restatement of the spectrogram front-end — TEST INFRASTRUCTURE ONLY (imported by tests/, never by asr_amd/).

Follows SpectrogramParser.parse_audio, asr_deepspeech/data/parsers/spectrogram_parser.py:45-60:
    n_fft = int(sr * window_size); hop = int(sr * window_stride)                      (:45-47)
    D = librosa.stft(y, n_fft=n_fft, hop_length=hop, win_length=n_fft, window=window)   (:49-51)
    spect = log1p(|D|) as float32                                                       (:52-55)
    if normalize: spect = (spect - spect.mean()) / spect.std()      (torch: unbiased)   (:56-60)

librosa is a THIRD-PARTY dependency absent from /root/reference and from this image (uv.lock pins librosa 0.11.0), so
`librosa.stft` is restated from its published algorithm: centre=True pads n_fft//2 samples on both sides (zeros — the
default `pad_mode="constant"` since librosa 0.10; "reflect" before that, kept selectable), frame t = padded[t*hop : t*hop+n_fft]
times scipy.signal.get_window(window, n_fft, fftbins=True), one-sided FFT, n_frames = 1 + len(y) // hop.
PARITY UNPINNED against librosa itself (cannot be imported here); tests/test_oracle_golden.py cross-checks this restatement
against two independent implementations that ARE present — torch.stft (documented librosa-compatible conventions) and
scipy.signal.stft — and the reference's own test properties (161 bins, finite, ~0 mean; tests/test_spectrogram_dataset.py:37-58).
"""
from __future__ import annotations

import numpy as np


def n_frames(n_samples: int, hop: int) -> int:
    return 1 + n_samples // hop


def stft_log_spectrogram(y, n_fft: int, hop: int, window: str = "hamming", pad_mode: str = "constant", normalize: bool = False) -> np.ndarray:
    """(n_fft/2+1, 1 + len(y)//hop) float64."""
    from scipy.signal import get_window
    y = np.asarray(y, dtype=np.float64)
    w = get_window(window, n_fft, fftbins=True).astype(np.float64)
    yp = np.pad(y, n_fft // 2, mode=pad_mode)
    T = n_frames(len(y), hop)
    frames = np.stack([yp[t * hop: t * hop + n_fft] * w for t in range(T)], axis=0)
    spect = np.log1p(np.abs(np.fft.rfft(frames, axis=1))).T
    if normalize:
        spect = (spect - spect.mean()) / spect.std(ddof=1)
    return spect


def batch_spectrogram(waves, n_fft: int, hop: int, window: str = "hamming", pad_mode: str = "constant", normalize: bool = False):
    """list of 1-D waveforms -> ((B,1,bins,Tmax) zero padded like _collate_fn (functional.py:18-30), frames list)."""
    specs = [stft_log_spectrogram(y, n_fft, hop, window, pad_mode, normalize) for y in waves]
    T = max(s.shape[1] for s in specs)
    out = np.zeros((len(specs), 1, n_fft // 2 + 1, T))
    for i, s in enumerate(specs):
        out[i, 0, :, :s.shape[1]] = s
    return out, [s.shape[1] for s in specs]
