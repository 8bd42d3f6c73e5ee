"""Shared helpers for parity tests (fixtures -> tensors, error metrics)."""
from __future__ import annotations

import json
import os

import numpy as np
import torch

import det  # tests/golden/det.py (on sys.path via conftest)

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
GRAD_STRIDE = 13
GRAD_FULL_MAX = 4096


def subsample(a):
    f = np.asarray(a).reshape(-1)
    return f if f.size <= GRAD_FULL_MAX else f[::GRAD_STRIDE]


def rel_l2(a, b) -> float:
    a = np.asarray(a, dtype=np.float64).reshape(-1)
    b = np.asarray(b, dtype=np.float64).reshape(-1)
    d = np.linalg.norm(a - b)
    n = np.linalg.norm(b)
    return float(d / n) if n > 0 else float(d)


def load_model_fixture(name):
    z = np.load(os.path.join(GOLDEN, f"model_{name}.npz"), allow_pickle=False)
    cfg = json.loads(str(z["cfg"]))
    return z, cfg


def model_inputs(cfg, device="cpu"):
    """(state_dict of torch tensors, inputs, targets, pct, target_sizes) regenerated from hashes."""
    shapes = det.state_shapes(cfg["rnn"], cfg["hidden"], cfg["layers"], cfg["classes"])
    w = det.model_state(shapes, base_seed=0)
    sd = {k: torch.from_numpy(np.asarray(v)).to(device) for k, v in w.items()}
    x, targets, pct, tsz = det.batch(len(cfg["t_ins"]), cfg["t_ins"], cfg["classes"], seed=1)
    return sd, torch.from_numpy(x), torch.from_numpy(targets), torch.from_numpy(pct), torch.from_numpy(tsz)


MODEL_FIXTURES = ["gru_h32_l2", "lstm_h24_l2", "gru_h48_l3", "lstm_h40_l3"]
