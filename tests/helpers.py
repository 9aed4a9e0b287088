"""Shared helpers for the parity tests."""
import os

import numpy as np
import torch

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_golden(name):
    return np.load(os.path.join(GOLDEN_DIR, name), allow_pickle=False)


def rel_err(a, b):
    """max |a-b| / max(|b|) -- the 'relative to the tensor's scale' error used for gradients."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    denom = max(float(np.max(np.abs(b))), 1e-30)
    return float(np.max(np.abs(a - b))) / denom


def sample_like_golden(t, G, stride):
    """Apply make_goldens.py's strided sampling of G-sized axes to tensor ``t``."""
    a = t.detach().cpu().numpy() if isinstance(t, torch.Tensor) else np.asarray(t)
    if stride == 1 or a.ndim != 2 or G not in a.shape:
        return a
    ax = a.shape.index(G)
    return np.take(a, np.arange(0, G, stride), axis=ax)
