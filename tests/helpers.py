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


def synthetic_study(n_cells=3000, n_genes=1003, seed=11):
    """A made-up study in the layout SingleCellDataset reads (data.py:41,63-66): ``{"obs": columns, "var": columns}``
    plus the uint8 cells x genes store.  Sparse counts, a few unknown / NaN / out-of-range annotations."""
    rng = np.random.RandomState(seed)
    store = (rng.poisson(0.6, size=(n_cells, n_genes)) * (rng.rand(n_cells, n_genes) < 0.5)).astype(np.uint8)
    age = rng.uniform(40, 95, n_cells)
    age[rng.rand(n_cells) < 0.03] = np.nan
    age[rng.rand(n_cells) < 0.02] = 1500.0
    obs = {
        "class": rng.choice(["OPC", "EN", "Astro"], n_cells, p=[0.7, 0.2, 0.1]),
        "include_training": (rng.rand(n_cells) < 0.95).astype(np.int64),
        "Dementia": rng.choice([0, 1, -1], n_cells, p=[0.45, 0.45, 0.1]),
        "BRAAK_CERAD": rng.choice([0, 1, 2, -1], n_cells, p=[0.35, 0.35, 0.2, 0.1]),
        "SubID": np.array([f"S{v:03d}" for v in rng.randint(0, 40, n_cells)]),
        "Age": age,
        "Sex": rng.choice(["M", "F", "U"], n_cells, p=[0.48, 0.48, 0.04]),
        "Brain_bank": rng.choice(["A", "B", "C"], n_cells),
    }
    var = {
        "gene_name": np.array([f"G{g}" for g in range(n_genes)]),
        "percent_cells": rng.uniform(0, 5, n_genes) * (rng.rand(n_genes) < 0.97),
        "gene_chrom": rng.choice(["1", "2", "X", "Y"], n_genes, p=[0.5, 0.4, 0.07, 0.03]),
    }
    var["gene_name"][5] = "MALAT1"
    return store, {"obs": obs, "var": var}


class HostOnlyStore:
    """Stands in for vae_amd.data.CountStore in the CPU tests of the dataset's host logic (no gather)."""

    def __init__(self, store):
        self.n_cells, self.n_genes = store.shape
        self.device = torch.device("cpu")
        self.upload = lambda a: torch.from_numpy(np.ascontiguousarray(a))
