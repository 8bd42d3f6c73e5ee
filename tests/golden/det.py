"""Platform-independent deterministic pseudo-random tensors (splitmix64 on the element index).

Golden fixtures store only *expected outputs*; inputs and weights are regenerated from
(seed, shape) with this integer-only generator, so they are bit-identical on every machine and
every torch/numpy version.  Values are multiples of 2^-24 in [0,1) -> exact in float32.
"""
from __future__ import annotations

import numpy as np

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix64(x: np.ndarray) -> np.ndarray:
    with np.errstate(over="ignore"):
        z = (x + np.uint64(0x9E3779B97F4A7C15)) & _M64
        z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M64
        z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M64
        z = z ^ (z >> np.uint64(31))
    return z


def uniform01(shape, seed: int) -> np.ndarray:
    n = int(np.prod(shape)) if len(shape) else 1
    with np.errstate(over="ignore"):
        idx = np.arange(n, dtype=np.uint64) + np.uint64(seed) * np.uint64(0x1000003D1)
    bits = _splitmix64(_splitmix64(idx))
    u = (bits >> np.uint64(40)).astype(np.float64) / float(1 << 24)
    return u.astype(np.float32).reshape(shape)


def uniform(shape, seed: int, lo: float, hi: float) -> np.ndarray:
    return (uniform01(shape, seed) * np.float32(hi - lo) + np.float32(lo)).astype(np.float32)


def unitvar(shape, seed: int) -> np.ndarray:
    """zero-mean unit-variance (uniform on [-sqrt3, sqrt3))."""
    r = 3.0 ** 0.5
    return uniform(shape, seed, -r, r)


def randint(shape, seed: int, lo: int, hi: int) -> np.ndarray:
    """integers in [lo, hi)."""
    u = uniform01(shape, seed).astype(np.float64)
    return (lo + np.floor(u * (hi - lo))).astype(np.int64).clip(lo, hi - 1)


def seed_of(name: str, base: int = 0) -> int:
    h = 1469598103934665603
    for ch in name.encode():
        h = ((h ^ ch) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
    return int((h ^ base) & 0x7FFFFFFF)


def model_state(shapes: dict, base_seed: int = 0) -> dict:
    """Deterministic weights for a state_dict shape manifest {key: shape}.  Weights/biases of
    conv/rnn/linear ~ U(-k, k), k = 1/sqrt(fan_in) (torch's default family); BN gamma ~ U(0.5,1.5),
    BN beta ~ U(-0.3,0.3); running_mean 0 / running_var 1; num_batches_tracked 0."""
    out = {}
    for key, shape in shapes.items():
        shape = tuple(shape)
        s = seed_of(key, base_seed)
        if key.endswith("num_batches_tracked"):
            out[key] = np.zeros((), dtype=np.int64)
        elif key.endswith("running_mean"):
            out[key] = np.zeros(shape, dtype=np.float32)
        elif key.endswith("running_var"):
            out[key] = np.ones(shape, dtype=np.float32)
        elif ("batch_norm" in key or key.startswith("conv.seq_module.1.") or key.startswith("conv.seq_module.4.")
              or key.startswith("fc.0.module.0.")):
            out[key] = uniform(shape, s, 0.5, 1.5) if key.endswith("weight") else uniform(shape, s, -0.3, 0.3)
        elif key.startswith("conv.seq_module."):
            fan_in = 1 * 41 * 11 if ".0." in key else 32 * 21 * 11
            k = 1.0 / fan_in ** 0.5
            out[key] = uniform(shape, s, -k, k)
        elif ".rnn." in key:
            hh_key = key.rsplit(".", 1)[0] + ".weight_hh_l0"
            hidden = shapes[hh_key][1]
            k = 1.0 / hidden ** 0.5
            out[key] = uniform(shape, s, -k, k)
        elif key.startswith("lookahead."):                      # unidirectional variant: depthwise Conv1d (H, 1, context)
            k = 1.0 / shape[-1] ** 0.5
            out[key] = uniform(shape, s, -k, k)
        elif key == "fc.0.module.1.weight":
            k = 1.0 / shape[1] ** 0.5
            out[key] = uniform(shape, s, -k, k)
        else:
            raise KeyError(key)
    return out


def state_shapes(rnn: str, hidden: int, layers: int, classes: int) -> dict:
    """state_dict key -> shape manifest of the reference DeepSpeech,
    bidirectional."""
    g = {"gru": 3, "lstm": 4}[rnn]
    sh = {}
    p = "conv.seq_module."
    sh[p + "0.weight"] = (32, 1, 41, 11)
    sh[p + "0.bias"] = (32,)
    for i in ("1", "3", "4"):
        if i == "3":
            sh[p + "3.weight"] = (32, 32, 21, 11)
            sh[p + "3.bias"] = (32,)
            continue
        for n in ("weight", "bias", "running_mean", "running_var"):
            sh[p + f"{i}.{n}"] = (32,)
        sh[p + f"{i}.num_batches_tracked"] = ()
    for l in range(layers):
        inp = 1312 if l == 0 else hidden
        if l > 0:
            for n in ("weight", "bias", "running_mean", "running_var"):
                sh[f"rnns.{l}.batch_norm.module.{n}"] = (hidden,)
            sh[f"rnns.{l}.batch_norm.module.num_batches_tracked"] = ()
        for sfx in ("", "_reverse"):
            sh[f"rnns.{l}.rnn.weight_ih_l0{sfx}"] = (g * hidden, inp)
            sh[f"rnns.{l}.rnn.weight_hh_l0{sfx}"] = (g * hidden, hidden)
            sh[f"rnns.{l}.rnn.bias_ih_l0{sfx}"] = (g * hidden,)
            sh[f"rnns.{l}.rnn.bias_hh_l0{sfx}"] = (g * hidden,)
    for n in ("weight", "bias", "running_mean", "running_var"):
        sh[f"fc.0.module.0.{n}"] = (hidden,)
    sh["fc.0.module.0.num_batches_tracked"] = ()
    sh["fc.0.module.1.weight"] = (classes, hidden)
    return sh


def batch(b: int, t_ins, classes: int, seed: int = 1):
    """Synthetic collated batch in the reference's input contract (functional.py:9-32):
    inputs (B,1,161,Tmax) zero beyond each T_b, lengths sorted descending, flat int32 targets
    in [1,C), U_b = max(1, T_b // 20), percentages T_b/Tmax in float32."""
    t_ins = sorted([int(t) for t in t_ins], reverse=True)
    assert len(t_ins) == b
    tmax = t_ins[0]
    x = unitvar((b, 1, 161, tmax), seed)
    for i, t in enumerate(t_ins):
        x[i, :, :, t:] = 0.0
    tgt_sizes = np.array([max(1, t // 20) for t in t_ins], dtype=np.int32)
    targets = randint((int(tgt_sizes.sum()),), seed + 1, 1, classes).astype(np.int32)
    pct = np.array([t / float(tmax) for t in t_ins], dtype=np.float64).astype(np.float32)
    return x, targets, pct, tgt_sizes


DECODE_CASES = (  # name, labels, B, T, quantisation levels (ties exercise the first-maximum rule), sizes
    ("small_ties", "_ABCDE ", 4, 50, 4, None),
    ("ragged", "_'ABCDEFGHIJKLMNOPQRSTUVWXYZ ", 3, 201, 0, [201, 120, 1]),
    ("blank_heavy", "_'ABCDEFGHIJKLMNOPQRSTUVWXYZ ", 5, 333, 3, [333, 332, 100, 64, 65]),
)


def decode_probs(name, b, t, c, levels):
    p = uniform01((b, t, c), seed_of("decode." + name))
    if levels:
        p = np.floor(p * levels) / levels
    if name == "blank_heavy":
        p[..., 0] = np.where(uniform01((b, t), seed_of("decode.blank." + name)) < 0.6, 2.0, p[..., 0])
    return p.astype(np.float32)
