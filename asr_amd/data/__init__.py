"""Data-side API surface kept by name: SpectrogramDataset, BucketingSampler,
DistributedBucketingSampler, AudioDataLoader, get_loader.  Host-side / I-O bound; not accelerated.

`SpectrogramDataset` reads the reference's manifest CSV (`audio_filepath`, `text`).  Audio decoding:
pre-computed spectrograms (`.npy` / `.pt`, shape (161, T)) are loaded as-is; `.wav` files go through a
numpy STFT restatement of data/parsers/spectrogram_parser.py:36-62 (n_fft = win = sr*window_size,
hop = sr*window_stride, centred zero-padded frames, log1p magnitude, per-utterance mean/std).
`GpuSpectrogramFrontEnd` does the same for a whole batch on the GPU (csrc/stft.hip).
Parity with librosa 0.11.0 itself is UNPINNED (librosa is not installable here); both are held to
oracle/stft_oracle.py, which is cross-checked against torch.stft and scipy.signal.stft.
"""
from __future__ import annotations

import math

import numpy as np
import torch
from torch.utils.data import DataLoader, Dataset
from torch.utils.data.sampler import Sampler

from ..functional import _collate_fn


def _stft_spectrogram(y: np.ndarray, sample_rate: int, window_size: float, window_stride: float, window: str = "hamming",
                      pad_mode: str = "constant"):
    """Host (numpy) restatement used by the per-item dataset path, like the reference's CPU librosa call.  Zero padding is
    librosa's default since 0.10 (the reference pins 0.11.0); pass pad_mode="reflect" for the pre-0.10 behaviour."""
    from scipy.signal import get_window
    n_fft = int(sample_rate * window_size)
    hop = int(sample_rate * window_stride)
    win = get_window(window, n_fft, fftbins=True).astype(np.float32)
    y = np.pad(y.astype(np.float32), n_fft // 2, mode=pad_mode)
    n_frames = 1 + (len(y) - n_fft) // hop
    frames = np.lib.stride_tricks.as_strided(y, shape=(n_frames, n_fft), strides=(y.strides[0] * hop, y.strides[0]))
    spect = np.abs(np.fft.rfft(frames * win, axis=1)).T.astype(np.float32)  # (n_fft/2+1, frames)
    return np.log1p(spect)


class GpuSpectrogramFrontEnd:
    """Batch spectrogram front-end on the GPU (csrc/stft.hip, `ds2_spectrogram_f32`): a list of 1-D waveforms in, the
    `_collate_fn` contract out — `(inputs (B,1,161,T) on the GPU, input_percentages (B,) float32)` — i.e. what
    SpectrogramParser.parse_audio (spectrogram_parser.py:36-62) + _collate_fn (functional.py:9-32) produce per batch, with
    the STFT, log1p, per-utterance mean/std and zero padding done in four kernels instead of per item in DataLoader workers."""

    def __init__(self, audio_conf, normalize=False, pad_mode="constant", device=None):
        self.n_fft = int(audio_conf.sample_rate * audio_conf.window_size)
        self.hop = int(audio_conf.sample_rate * audio_conf.window_stride)
        self.window, self.normalize, self.pad_mode, self.device = audio_conf.window, normalize, pad_mode, device

    def __call__(self, waves):
        from .. import ops
        from ..device import resolve_device
        dev = torch.device(self.device) if self.device is not None else resolve_device("auto")
        n = [int(len(w)) for w in waves]
        batch = torch.zeros(len(waves), max(n), dtype=torch.float32)
        for i, w in enumerate(waves):
            batch[i, :n[i]] = torch.as_tensor(w, dtype=torch.float32)
        spect, frames = ops.spectrogram(batch.to(dev), torch.tensor(n), self.n_fft, self.hop, self.window, self.pad_mode, self.normalize)
        return spect, frames.float() / float(spect.size(3))


class SpectrogramDataset(Dataset):
    def __init__(self, audio_conf, manifest_filepath, labels, normalize=False, spec_augment=False, caching=False):
        import pandas as pd
        self.df = pd.read_csv(manifest_filepath)
        self.size = len(self.df)
        if isinstance(labels, str):
            labels = dict([(v, k) for k, v in pd.read_csv(labels).to_dict()["label"].items()])
        self.labels_map = labels
        self.audio_conf, self.normalize, self.caching = audio_conf, normalize, caching
        self._cache = {}

    def parse_audio(self, path):
        if path.endswith(".npy"):
            spect = torch.from_numpy(np.load(path)).float()
        elif path.endswith(".pt"):
            spect = torch.load(path).float()
        else:
            from scipy.io import wavfile
            sr, y = wavfile.read(path)
            if y.dtype.kind == "i":
                y = y.astype(np.float32) / float(np.iinfo(y.dtype).max)
            if y.ndim > 1:
                y = y.mean(axis=1)
            assert sr == self.audio_conf.sample_rate, f"expected {self.audio_conf.sample_rate} Hz audio"
            spect = torch.from_numpy(_stft_spectrogram(y, sr, self.audio_conf.window_size, self.audio_conf.window_stride,
                                                       self.audio_conf.window))
        if self.normalize:
            spect = (spect - spect.mean()) / spect.std()
        return spect

    def parse_transcript(self, transcript):
        """spectrogram_dataset.py:70-73: unknown chars and the index-0 (blank) label are dropped."""
        transcript = transcript.replace("\n", "")
        return list(filter(None, [self.labels_map.get(x) for x in list(transcript)]))

    def __getitem__(self, index):
        row = self.df.iloc[index]
        if self.caching and index in self._cache:
            return self._cache[index]
        item = (self.parse_audio(row.audio_filepath), self.parse_transcript(row.text))
        if self.caching:
            self._cache[index] = item
        return item

    def __len__(self):
        return self.size


class BucketingSampler(Sampler):
    """data/samplers/bucketing_sampler.py:5-25: consecutive manifest rows form a bin (= batch)."""

    def __init__(self, data_source, batch_size=1):
        super().__init__()
        self.data_source = data_source
        ids = list(range(len(data_source)))
        self.bins = [ids[i:i + batch_size] for i in range(0, len(ids), batch_size)]

    def __iter__(self):
        for ids in self.bins:
            np.random.shuffle(ids)
            yield ids

    def __len__(self):
        return len(self.bins)

    def shuffle(self, epoch=None):
        np.random.shuffle(self.bins)


class DistributedBucketingSampler(Sampler):
    """data/samplers/distributed_bucketing_sampler.py:8-44 (dead code in the reference, the DP
    partition rule here): rank r takes bins[r::world], wrap-padded to a multiple of world."""

    def __init__(self, data_source, batch_size=1, num_replicas=None, rank=None):
        super().__init__()
        if num_replicas is None or rank is None:
            import torch.distributed as dist
            num_replicas = dist.get_world_size() if num_replicas is None else num_replicas
            rank = dist.get_rank() if rank is None else rank
        self.data_source = data_source
        self.ids = list(range(len(data_source)))
        self.batch_size = batch_size
        self.bins = [self.ids[i:i + batch_size] for i in range(0, len(self.ids), batch_size)]
        self.num_replicas, self.rank = num_replicas, rank
        self.num_samples = int(math.ceil(len(self.bins) * 1.0 / self.num_replicas))
        self.total_size = self.num_samples * self.num_replicas

    def __iter__(self):
        bins = self.bins + self.bins[: (self.total_size - len(self.bins))]
        assert len(bins) == self.total_size
        return iter(bins[self.rank::self.num_replicas])

    def __len__(self):
        return self.num_samples

    def shuffle(self, epoch):
        g = torch.Generator()
        g.manual_seed(epoch)
        order = torch.randperm(len(self.bins), generator=g).tolist()
        self.bins = [self.bins[i] for i in order]


class AudioDataLoader(DataLoader):
    """data/loaders/audio_data_loader.py:7-13."""

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.collate_fn = _collate_fn


def get_loader(audio_conf, labels, manifest, batch_size, num_workers, caching=False):
    """data/loaders/functional.py:6-24."""
    dataset = SpectrogramDataset(audio_conf=audio_conf, manifest_filepath=manifest, labels=labels, normalize=True,
                                 spec_augment=getattr(audio_conf, "spec_augment", False), caching=caching)
    sampler = BucketingSampler(dataset, batch_size=batch_size)
    loader = AudioDataLoader(dataset, num_workers=num_workers, batch_sampler=sampler)
    sampler.shuffle()
    return loader, sampler
