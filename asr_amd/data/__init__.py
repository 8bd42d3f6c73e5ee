"""Data-side API surface kept by name: SpectrogramDataset, BucketingSampler,
DistributedBucketingSampler, AudioDataLoader, get_loader; plus the true length-bucketing samplers of SURVEY §8(f)4
(LengthBucketingSampler, DistributedLengthBucketingSampler).  Host-side / I-O bound; not accelerated.

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
        # spectrogram_parser.py:29-44 / :64-80: noise injection, tempo/gain perturbation and SpecAugment are host-side augmentations of
        # the reference's parser that this loader does not implement — say so instead of training silently without them
        unsupported = [k for k, on in (("noise_dir", getattr(audio_conf, "noise_dir", None) is not None),
                                       ("speed_volume_perturb", bool(getattr(audio_conf, "speed_volume_perturb", False))),
                                       ("spec_augment", bool(spec_augment))) if on]
        if unsupported:
            import warnings
            warnings.warn("asr_amd.data.SpectrogramDataset: augmentation(s) requested by audio_conf but not implemented here, ignored: "
                          + ", ".join(unsupported))

    def parse_audio(self, path):
        if path.endswith(".npy"):
            spect = torch.from_numpy(np.load(path)).float()
        elif path.endswith(".pt"):
            spect = torch.load(path).float()
        else:
            from scipy.io import wavfile
            sr, y = wavfile.read(path)
            if y.dtype.kind == "i":
                # soundfile (audio/functional.py:11) scales integer PCM by 2^(bits-1): int16 / 32768
                y = y.astype(np.float32) / float(2 ** (8 * y.dtype.itemsize - 1))
            elif y.dtype.kind == "u":
                y = (y.astype(np.float32) - 128.0) / 128.0      # 8-bit WAV is unsigned
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


def _durations_of(data_source, durations=None):
    """Per-item lengths for bucketing: explicit sequence, else the manifest's `duration` column (etl/jsut_dataset.py:36-42,
    etl/librispeech_dataset.py:100-103 write it), else the text length as a proxy."""
    if durations is not None:
        d = np.asarray(list(durations), dtype=np.float64)
    else:
        df = getattr(data_source, "df", None)
        if df is not None and "duration" in df.columns:
            d = df["duration"].to_numpy(dtype=np.float64)
        elif df is not None and "text_size" in df.columns:
            d = df["text_size"].to_numpy(dtype=np.float64)
        else:
            raise ValueError("length bucketing needs per-item durations: pass `durations=` or a manifest with a `duration` column")
    if len(d) != len(data_source):
        raise ValueError(f"{len(d)} durations for {len(data_source)} items")
    return d


def _item_order(self, data_source, durations, descending, order):
    """Item ids in binning order: by duration (stable; ties by manifest index) or as the manifest lists them."""
    if order == "manifest":
        self.durations = None
        return list(range(len(data_source)))
    if order != "duration":
        raise ValueError(f"order={order!r}: expected duration or manifest")
    self.durations = _durations_of(data_source, durations)
    return np.argsort(-self.durations if descending else self.durations, kind="stable").tolist()


def _full_bins(order, batch_size, partial):
    """Consecutive runs of `batch_size` ids of a length-sorted order.  A short last run is kept as it is ("keep": the reference's
    BucketingSampler behaviour, bucketing_sampler.py:13-14), dropped ("drop") or topped up with its nearest-in-length neighbours — the
    ids just before it, which then occur twice in the epoch ("fill")."""
    if partial not in ("keep", "drop", "fill"):
        raise ValueError(f"partial={partial!r}: expected keep, drop or fill")
    bins = [order[i:i + batch_size] for i in range(0, len(order), batch_size)]
    if bins and len(bins[-1]) < batch_size and len(order) >= batch_size:
        if partial == "drop":
            bins.pop()
        elif partial == "fill":
            need = batch_size - len(bins[-1])
            bins[-1] = order[len(order) - len(bins[-1]) - need:len(order) - len(bins[-1])] + bins[-1]
    return bins


class LengthBucketingSampler(Sampler):
    """True length bucketing (BASELINE configs[3] "bucketed sampler" / configs[4] "length-sorted batching").

    The reference's BucketingSampler (data/samplers/bucketing_sampler.py:5-25) only ASSUMES a manifest "in order of size" (its ETL merely
    bounds the lengths, etl/__main__.py:48); this one establishes the order: items are sorted by duration (stable, ties by manifest
    index), consecutive runs of `batch_size` form the bins, and the interface is the reference's — `__iter__` yields one bin per batch
    (ids shuffled inside the bin like the reference; `_collate_fn` re-sorts a batch by length anyway), `__len__` = number of bins,
    `shuffle()` permutes the bin ORDER (batches stay homogeneous in length, epochs see them in a different order)."""

    def __init__(self, data_source, batch_size=1, durations=None, descending=False, partial="keep", order="duration"):
        super().__init__()
        self.data_source = data_source
        self.batch_size = int(batch_size)
        self.bins = _full_bins(_item_order(self, data_source, durations, descending, order), self.batch_size, partial)

    def __iter__(self):
        for ids in self.bins:
            np.random.shuffle(ids)
            yield ids

    def __len__(self):
        return len(self.bins)

    def shuffle(self, epoch=None):
        if epoch is None:
            np.random.shuffle(self.bins)                 # the reference's call (bucketing_sampler.py:24-25)
        else:
            g = torch.Generator()
            g.manual_seed(int(epoch))
            self.bins = [self.bins[i] for i in torch.randperm(len(self.bins), generator=g).tolist()]

    def bin_spread(self):
        """max - min duration inside each bin (what bucketing minimises; used by the tests and the bench)."""
        return [float(self.durations[b].max() - self.durations[b].min()) for b in self.bins]


class BucketingSampler(LengthBucketingSampler):
    """The reference's sampler of this name (data/samplers/bucketing_sampler.py:5-25: bins of consecutive manifest rows, "assuming they are
    in order of size") = the length-bucketing sampler with the manifest order taken as given.  Bins, in-bin shuffle and `shuffle()` are
    held to the reference's own output in tests/golden/data_formats.json."""

    def __init__(self, data_source, batch_size=1):
        super().__init__(data_source, batch_size=batch_size, order="manifest")


class DistributedLengthBucketingSampler(Sampler):
    """Length bucketing for one-process-per-GPU data parallelism: the partition rule is the reference's (rank r takes every
    `num_replicas`-th bin starting at r, distributed_bucketing_sampler.py:22-34), applied to LENGTH-SORTED bins, and the epoch shuffle
    moves whole ROUNDS (the `num_replicas` bins that the ranks process concurrently) instead of single bins — so at every step all ranks
    hold batches that are neighbours in length, and the gradient all-reduce does not wait for a straggler with a much longer T
    ("for C5 sort by length first so concurrent ranks get similar T").  The tail is padded to a whole round with the
    bins just before it (nearest in length), where the reference wraps around to the first bins (which here would put the SHORTEST
    batches next to the LONGEST ones in the last round).

    `order="manifest"`, `pad="wrap"`, `shuffle_unit="bin"` select the reference's own behaviour instead (DistributedBucketingSampler below):
    manifest order, wrap-around padding applied AFTER the shuffle, single bins permuted."""

    _fill_logged = False

    def __init__(self, data_source, batch_size=1, num_replicas=None, rank=None, durations=None, descending=False, partial="keep",
                 order="duration", pad="nearest", shuffle_unit="round"):
        super().__init__()
        if num_replicas is None:
            num_replicas = torch.distributed.get_world_size()
        if rank is None:
            rank = torch.distributed.get_rank()
        if pad not in ("nearest", "wrap") or shuffle_unit not in ("round", "bin"):
            raise ValueError(f"pad={pad!r} / shuffle_unit={shuffle_unit!r}: expected nearest | wrap and round | bin")
        self.data_source, self.batch_size = data_source, int(batch_size)
        self.num_replicas, self.rank = int(num_replicas), int(rank)
        self.pad, self.shuffle_unit = pad, shuffle_unit
        self.ids = _item_order(self, data_source, durations, descending, order)
        # partial="keep" (default, the reference sampler's behaviour: the short last bin stays short, epochs are composed exactly as with
        # data/samplers/distributed_bucketing_sampler.py).  partial="fill" tops the short bin up with the ids just before it, so that every
        # rank of every round runs the SAME batch size — gradients are averaged unweighted, and a rank whose B is not a multiple of 8 leaves
        # the packed bf16 fast path and straggles in the all-reduce; it duplicates samples within an epoch (logged once), so it is opt-in.
        self.bins = _full_bins(self.ids, self.batch_size, partial)
        if partial == "fill" and len(self.ids) % self.batch_size and not DistributedLengthBucketingSampler._fill_logged:
            DistributedLengthBucketingSampler._fill_logged = True
            print(f"[asr_amd] DistributedLengthBucketingSampler(partial='fill'): the last bin is topped up with "
                  f"{self.batch_size - len(self.ids) % self.batch_size} duplicate sample(s) per epoch", flush=True)
        self.num_samples = int(math.ceil(len(self.bins) / self.num_replicas))
        self.total_size = self.num_samples * self.num_replicas
        self._round_order = None                                   # shuffle_unit="round": permutation of the rounds, set by shuffle()

    def _padded(self):
        """The bins of one epoch, padded to a whole number of rounds."""
        pad = self.total_size - len(self.bins)
        if self.pad == "wrap":                                     # the reference: the epoch's first bins again
            bins = self.bins + self.bins[:pad]
            assert len(bins) == self.total_size                    # (its own assertion: fails when there are fewer bins than padding)
            return bins
        if not pad:
            return self.bins
        src = self.bins[-(pad + 1):-1] if len(self.bins) > pad else (self.bins * (pad // max(len(self.bins), 1) + 1))[:pad]
        return self.bins + [list(b) for b in src]

    @property
    def rounds(self):
        bins = self._padded()
        rounds = [bins[i:i + self.num_replicas] for i in range(0, self.total_size, self.num_replicas)]
        return rounds if self._round_order is None else [rounds[i] for i in self._round_order]

    def __iter__(self):
        return iter([r[self.rank] for r in self.rounds])

    def __len__(self):
        return self.num_samples

    def shuffle(self, epoch):
        g = torch.Generator()
        g.manual_seed(int(epoch))
        if self.shuffle_unit == "bin":
            self.bins = [self.bins[i] for i in torch.randperm(len(self.bins), generator=g).tolist()]
        else:
            prev = self._round_order if self._round_order is not None else list(range(self.num_samples))
            self._round_order = [prev[i] for i in torch.randperm(self.num_samples, generator=g).tolist()]


class DistributedBucketingSampler(DistributedLengthBucketingSampler):
    """The reference's sampler of this name (data/samplers/distributed_bucketing_sampler.py:8-44; dead code there, the SURVEY §8(e) partition
    rule here): bins of consecutive manifest rows, rank r takes bins[r::world] of the bin list wrap-padded to a multiple of world, `shuffle(epoch)`
    permutes the bins with a generator seeded by the epoch.  Same attributes (`ids`, `bins`, `num_replicas`, `rank`, `num_samples`,
    `total_size`); per-rank bins for world 1 / 2 / 4 / 8 and the shuffles are held to the reference's own output (tests/golden/data_formats.json)."""

    def __init__(self, data_source, batch_size=1, num_replicas=None, rank=None):
        super().__init__(data_source, batch_size=batch_size, num_replicas=num_replicas, rank=rank, order="manifest", pad="wrap",
                         shuffle_unit="bin")


def write_manifest(records, path):
    """Manifest CSV in the layout the reference's ETL writes (etl/jsut_dataset.py:36-42 + etl/__main__.py:55: DataFrame.to_csv(index=False)
    with the columns audio_filepath, duration, fq, text, text_size).  `records`: iterable of (audio_filepath, duration, fq, text)."""
    import pandas as pd
    rows = [(str(f), float(d), int(fq), str(t), len(str(t))) for f, d, fq, t in records]
    pd.DataFrame.from_records(rows, columns=["audio_filepath", "duration", "fq", "text", "text_size"]).to_csv(path, index=False)


def export_labels(texts, path):
    """labels.csv as the reference writes it (etl/jsut_dataset.py:56-60): one `label` column holding every character of the corpus; the
    reference emits the set in hash order, this writes it sorted (index 0 = CTC blank is whatever row comes first, as there)."""
    import pandas as pd
    chars = set()
    for t in texts:
        chars |= set(t)
    pd.DataFrame.from_records([(c,) for c in sorted(chars)], columns=["label"]).to_csv(path, index=False)


def clean_jsut_text(line):
    """`key:text` line of a JSUT transcript_utf8.txt -> (key, text) with spaces, newlines and the two Japanese punctuation marks removed
    (etl/jsut_dataset.py:46-50)."""
    k, v = line.split(":")
    for ch in (" ", "\n", "\u3001", "\u3002"):
        v = v.replace(ch, "")
    return k, v


class AudioDataLoader(DataLoader):
    """data/loaders/audio_data_loader.py:7-13."""

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.collate_fn = _collate_fn


def get_loader(audio_conf, labels, manifest, batch_size, num_workers, caching=False, length_bucketing=False):
    """data/loaders/functional.py:6-24.  `length_bucketing=True` (not in the reference) sorts the manifest by its `duration` column
    before binning (LengthBucketingSampler; the distributed variant when torch.distributed is initialised)."""
    dataset = SpectrogramDataset(audio_conf=audio_conf, manifest_filepath=manifest, labels=labels, normalize=True,
                                 spec_augment=getattr(audio_conf, "spec_augment", False), caching=caching)
    if length_bucketing and torch.distributed.is_available() and torch.distributed.is_initialized():
        sampler = DistributedLengthBucketingSampler(dataset, batch_size=batch_size)
        loader = AudioDataLoader(dataset, num_workers=num_workers, batch_sampler=sampler)
        sampler.shuffle(0)
        return loader, sampler
    sampler = (LengthBucketingSampler if length_bucketing else BucketingSampler)(dataset, batch_size=batch_size)
    loader = AudioDataLoader(dataset, num_workers=num_workers, batch_sampler=sampler)
    sampler.shuffle()
    return loader, sampler
