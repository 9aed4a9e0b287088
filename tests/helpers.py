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


# ---- the library's documented dropout stream (include/vae_amd.h "dropout"), restated on the host ----
def philox4(seed: int, step: int, site: int, elem: np.ndarray) -> np.ndarray:
    """The four output words (N, 4) of Philox4x32-10 with key (seed lo, seed hi) and counter
    (elem lo, elem hi, site, step)."""
    elem = np.asarray(elem, dtype=np.uint64)
    m32 = np.uint64(0xFFFFFFFF)
    c0, c1 = elem & m32, elem >> np.uint64(32)
    c2 = np.full_like(c0, site)
    c3 = np.full_like(c0, step)
    k0, k1 = np.uint64(seed & 0xFFFFFFFF), np.uint64((seed >> 32) & 0xFFFFFFFF)
    for _ in range(10):
        p0 = np.uint64(0xD2511F53) * c0
        p1 = np.uint64(0xCD9E8D57) * c2
        c0, c1, c2, c3 = ((p1 >> np.uint64(32)) ^ c1 ^ k0) & m32, p1 & m32, ((p0 >> np.uint64(32)) ^ c3 ^ k1) & m32, p0 & m32
        k0 = (k0 + np.uint64(0x9E3779B9)) & m32
        k1 = (k1 + np.uint64(0xBB67AE85)) & m32
    return np.stack([c0, c1, c2, c3], axis=-1).astype(np.uint32)


def philox_word0(seed: int, step: int, site: int, elem: np.ndarray) -> np.ndarray:
    return philox4(seed, step, site, elem)[..., 0]


def gene_dropout_keep(threshold: np.ndarray, seed: int, step: int, B: int) -> np.ndarray:
    """(B, G) bool keep mask of GeneDropout on the z-encoder input (include/vae_amd.h, vae_gene_dropout):
    site 1, counter element b * (G/4) + g/4, gene g kept iff word g%4 < threshold[g]."""
    G = threshold.shape[0]
    words = philox4(seed, step, 1, np.arange(B * (G // 4), dtype=np.uint64)).reshape(B, G)
    return words < threshold.astype(np.uint32)[None, :]


def dropout_multiplier(p: float, seed: int, step: int, site: int, B: int, W: int) -> np.ndarray:
    """(B, W) float32: 1 / (1 - p) where the element is kept, 0 where it is dropped."""
    elem = np.arange(B * W, dtype=np.uint64)
    keep = philox_word0(seed, step, site, elem) >= np.uint32(min(int(np.float32(p) * 4294967296.0), 4294967295))
    scale = np.float32(1.0 / (1.0 - float(np.float32(p))))
    return np.where(keep, scale, np.float32(0)).astype(np.float32).reshape(B, W)
