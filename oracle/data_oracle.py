"""CPU oracle for the input side of the training step.

TEST INFRASTRUCTURE ONLY: nothing under ``vae_amd/`` imports this file (see ``oracle/vae_oracle.py``).

A numpy restatement -- written from the reference's behaviour, not copied -- of what
``SingleCellDataset`` (``/root/reference/src/data.py``) does to build one batch: the uint8 row gather with the
gene subset, the balanced batch sampler, mixup, the label / mask tables and the library-size prior.  Every
random draw goes through the same ``np.random`` global-state calls, in the same order, as the reference makes,
so that under one ``np.random.seed`` the index streams are the reference's.  Plain loops, small cases only.

PARITY UNPINNED against the reference itself: ``data.py`` imports ``pytorch_lightning``, which is not
installed here, the reference ships no tests or data fixtures for it, and its metadata files are pickles of
the authors' study that are not part of the repository.  What pins this file is that it consists of the
reference's own numpy calls (cited per function); the GPU path is then held bit-exact to it.
"""
from __future__ import annotations

from typing import Any, Dict, List, Optional, Sequence

import numpy as np

# the two-column group definitions of data.py:109-203: scheme -> (column, values of that column)
GROUP_SCHEMES = {
    "cd": ("CERAD", range(1, 5)),        # data.py:114-125 (group key (c - 1) * 10 + d)
    "bd": ("BRAAK_AD", range(7)),        # data.py:127-139
    "bd5": ("BRAAK_AD5", range(6)),      # data.py:176-188
    "bcd": ("BRAAK_CERAD", range(9)),    # data.py:191-203
}


def gene_vals_batch(store: np.ndarray, cell_idx: np.ndarray, batch_idx: Sequence[int], gene_idx: np.ndarray) -> np.ndarray:
    """data.py:406-415: row ``cell_idx[i]`` of the uint8 store, genes ``gene_idx``, as float32."""
    out = np.zeros((len(batch_idx), len(gene_idx)), dtype=np.float32)
    for n, i in enumerate(batch_idx):
        out[n, :] = store[cell_idx[i]][gene_idx].astype(np.float32)
    return out


def sample_beta(size: int, concentration: float = 0.25) -> np.ndarray:
    """data.py:425-429: Beta(c, c) as a ratio of two gamma draws (global numpy state)."""
    g1 = np.random.gamma(concentration, size=size)
    g2 = np.random.gamma(concentration, size=size)
    return g1 / (g1 + g2)


def mixup(gene_vals: np.ndarray, cell_vals: np.ndarray, cell_mask: np.ndarray, cell_per_group: Optional[int]):
    """data.py:457-477.  Returns (new_gene_vals, new_cell_vals, new_cell_mask, alpha, partners)."""
    N = gene_vals.shape[0]
    alpha = sample_beta(N)
    new_g, new_v, new_m = np.zeros_like(gene_vals), np.zeros_like(cell_vals), np.zeros_like(cell_mask)
    partners = np.zeros(N, dtype=np.int64)
    for i in range(N):
        if cell_per_group is not None:
            k = i // cell_per_group
            candidates = list(range(k * cell_per_group, (k + 1) * cell_per_group))
        else:
            candidates = list(range(N))
        j = np.random.choice(candidates)
        partners[i] = j
        new_g[i, :] = alpha[i] * gene_vals[i, :] + (1 - alpha[i]) * gene_vals[j, :]
        new_v[i, :] = alpha[i] * cell_vals[i, :] + (1 - alpha[i]) * cell_vals[j, :]
        new_m[i, :] = alpha[i] * cell_mask[i, :] + (1 - alpha[i]) * cell_mask[j, :]
    return new_g, new_v, new_m, alpha, partners


def cutmix(gene_vals: np.ndarray, cell_vals: np.ndarray, cell_mask: np.ndarray, cell_per_group: Optional[int], n_genes: int):
    """data.py:431-454.  Returns (new_gene_vals, new_cell_vals, new_cell_mask, alpha, partners, take) with take[i, g] = 1
    where gene g of cell i was taken from its partner."""
    N = gene_vals.shape[0]
    alpha = sample_beta(N)
    new_g, new_v, new_m = np.zeros_like(gene_vals), np.zeros_like(cell_vals), np.zeros_like(cell_mask)
    partners = np.zeros(N, dtype=np.int64)
    take = np.zeros(gene_vals.shape, dtype=np.uint8)
    for i in range(N):
        if cell_per_group is not None:
            k = i // cell_per_group
            candidates = list(range(k * cell_per_group, (k + 1) * cell_per_group))
        else:
            candidates = list(range(N))
        j = np.random.choice(candidates)
        partners[i] = j
        new_g[i, :] += gene_vals[i, :]
        idx_mix = np.random.choice(n_genes, int(alpha[i] * n_genes), replace=False)
        new_g[i, idx_mix] = gene_vals[j, idx_mix]
        take[i, idx_mix] = 1
        new_v[i, :] = alpha[i] * cell_vals[i, :] + (1 - alpha[i]) * cell_vals[j, :]
        new_m[i, :] = alpha[i] * cell_mask[i, :] + (1 - alpha[i]) * cell_mask[j, :]
    return new_g, new_v, new_m, alpha, partners, take


def balanced_batch(group_idx: Dict[int, List[int]], cell_per_group: int):
    """data.py:480-495: ``cell_per_group`` cells of every group, without replacement, groups in dict order."""
    batch_idx: List[int] = []
    groups: List[int] = []
    for k, v in group_idx.items():
        idx = np.random.choice(v, size=cell_per_group, replace=False)
        batch_idx += idx.tolist()
        groups += [k] * cell_per_group
    return batch_idx, groups


def define_groups(obs: Dict[str, np.ndarray], scheme: str, batch_size: int, min_cells: int = 250):
    """data.py:109-215 for the two-column schemes: groups (value of the staging column) x (Dementia 0/1), key
    ``10 * b + d`` (``cd``: ``10 * (c - 1) + d``), group 999 = either column negative, groups under
    ``min_cells`` cells dropped, ``cell_per_group = batch_size // n_groups``."""
    col, values = GROUP_SCHEMES[scheme]
    groups: Dict[int, List[int]] = {}
    for b in values:
        for d in range(2):
            cond = (obs["Dementia"] == d) * (obs[col] == b)
            key = (b - 1) * 10 + d if scheme == "cd" else b * 10 + d
            groups[key] = np.where(cond)[0].tolist()
    cond = (obs["Dementia"] < 0) + (obs[col] < 0)
    groups[999] = np.where(cond)[0].tolist()
    groups = {k: v for k, v in groups.items() if len(v) >= min_cells}
    return groups, batch_size // len(groups)


def restrict_samples(obs: Dict[str, np.ndarray], cell_idx: np.ndarray, restrictions: Optional[Dict[str, Any]]):
    """data.py:247-288 (generic part): cells of ``cell_idx`` with ``include_training == 1`` that satisfy every
    restriction (a scalar: equality; a list: any of).  Returns the surviving row numbers."""
    cond = np.zeros(len(obs["class"]), dtype=np.uint8)
    cond[cell_idx] = 1
    cond *= obs["include_training"] == 1
    if restrictions is not None:
        for k, v in restrictions.items():
            if isinstance(v, list):
                cond *= np.sum(np.stack([obs[k] == v1 for v1 in v]), axis=0).astype(np.uint8)
            else:
                cond *= obs[k] == v
    return np.where(cond)[0]


def cell_prop_arrays(obs: Dict[str, np.ndarray], cell_properties: Dict[str, Any], max_cell_prop_val: float = 999):
    """data.py:350-390: (labels, mask) float32 (n, P); continuous: (v - mean) / std, out of range / NaN -> -100 and
    mask 0; discrete: index into ``values``, unknown -> -100 and mask 0."""
    n = len(next(iter(obs.values())))
    P = len(cell_properties)
    labels = np.zeros((n, P), dtype=np.float32)
    mask = np.ones((n, P), dtype=np.float32)
    for n0 in range(n):
        for n1, (k, prop) in enumerate(cell_properties.items()):
            v = obs[k][n0]
            if not prop["discrete"]:
                if v > max_cell_prop_val or v < -max_cell_prop_val or np.isnan(v):
                    labels[n0, n1] = -100
                    mask[n0, n1] = 0.0
                else:
                    labels[n0, n1] = (v - prop["mean"]) / prop["std"]
            else:
                idx = np.where(v == np.array(prop["values"]))[0]
                if len(idx) == 0:
                    labels[n0, n1] = -100
                    mask[n0, n1] = 0.0
                else:
                    labels[n0, n1] = idx[0]
    return labels, mask


def batch_prop_arrays(obs: Dict[str, np.ndarray], batch_properties: Dict[str, Any]):
    """data.py:327-348: (batch_labels, batch_mask) int64 (n, Pb); unknown value -> -100 and mask 0."""
    n = len(next(iter(obs.values())))
    P = len(batch_properties)
    labels = np.zeros((n, P), dtype=np.int64)
    mask = np.ones((n, P), dtype=np.int64)
    for n0 in range(n):
        for n1, (k, prop) in enumerate(batch_properties.items()):
            idx = np.where(obs[k][n0] == np.array(prop["values"]))[0]
            if len(idx) == 0:
                labels[n0, n1] = -100
                mask[n0, n1] = 0
            else:
                labels[n0, n1] = idx[0]
    return labels, mask


def cell_prop_info(obs: Dict[str, np.ndarray], cell_properties: Dict[str, Any], max_cell_prop_val: float = 999):
    """DataModule._get_cell_prop_info (data.py:594-659): fills ``mean`` / ``std`` of continuous properties (over
    values in [0, max)), ``values`` / ``freq`` of discrete ones (freq = count / mean count)."""
    out = {k: dict(v) for k, v in cell_properties.items()}
    for k, prop in out.items():
        vals = np.array(obs[k])
        if "freq" in prop or "mean" in prop:
            continue
        if not prop["discrete"]:
            idx = [n for n, cv in enumerate(vals) if cv >= 0 and cv < max_cell_prop_val]
            prop["mean"] = np.mean(vals[idx])
            prop["std"] = np.std(vals[idx])
        elif prop["values"] is None:
            uniq, counts = np.unique(vals, return_counts=True)
            idx = [n for n, u in enumerate(uniq) if (isinstance(u, str) or (u >= 0 and u < max_cell_prop_val))]
            prop["values"] = uniq[idx]
            prop["freq"] = counts[idx] / np.mean(counts[idx])
        else:
            uniq, counts = np.unique(vals, return_counts=True)
            idx = [n for n, u in enumerate(uniq) if u in prop["values"]]
            prop["freq"] = counts[idx] / np.mean(counts[idx])
    return out


def library_size_stats(store: np.ndarray, cell_idx: np.ndarray, gene_idx: np.ndarray, N: int = 200_000):
    """data.py:233-244: mean and variance of log1p(float32 row sum over the gene subset) of the first N cells."""
    idx = cell_idx[:N] if len(cell_idx) > N else cell_idx
    counts = []
    for n in idx:
        data = store[n][gene_idx].astype(np.float32)
        counts.append(np.log1p(np.sum(data, axis=-1)))
    return np.mean(counts), np.var(counts)
