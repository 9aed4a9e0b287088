"""Input side of the training step: the reference's ``SingleCellDataset`` / ``DataModule`` (src/data.py) with the
count matrix resident in HBM.

The reference opens one ``np.memmap`` per cell on the CPU, casts to float32 and ships 4 bytes per count to the GPU
(data.py:406-415).  Here the whole uint8 cells x genes file (create_dataset.py:131-142) is copied to the device
ONCE -- 5 M cells x 30 k genes is 150 GB, an MI355X holds 288 GB -- and a batch is a HIP gather of its rows
(``vae_gather_counts``), optionally followed by mixup on the device (``vae_mixup_counts``, bit-identical to the
reference's numpy arithmetic).  Per step the host only draws indices.

What is mirrored: ``__getitem__`` -> the 8-tuple ``(gene_vals, cell_prop_vals, cell_mask, batch_labels, batch_mask,
cell_idx, group_idx, subject)`` (data.py:497-521), the balanced batch sampler (data.py:480-495, same ``np.random``
call sequence, so the same indices under the same seed), mixup and cutmix (data.py:431-477, same draws), the label / mask tables
(data.py:327-390), sample restriction (data.py:247-288), the two-column group schemes (data.py:109-215),
``library_size_stats`` (data.py:233-244), and ``DataModule`` with ``_get_cell_prop_info`` / ``setup`` /
``train_dataloader`` / ``val_dataloader`` (data.py:524-729; torch's own ``BatchSampler(RandomSampler)``).
Not mirrored: the study-specific ``bd_augmented`` / ``bd_special`` group
schemes and the polygenic-score restrictions.  No CPU fallback: without the HIP library this module raises.
"""
from __future__ import annotations

import pickle
from typing import Any, Dict, List, Optional, Sequence, Union

import numpy as np
import torch
from torch.utils.data import BatchSampler, RandomSampler, SequentialSampler

from . import ops

# staging column and its values per two-column group scheme (data.py:114-203)
_GROUP_SCHEMES = {"cd": ("CERAD", range(1, 5)), "bd": ("BRAAK_AD", range(7)), "bd5": ("BRAAK_AD5", range(6)),
                  "bcd": ("BRAAK_CERAD", range(9))}


class _Uploader:
    """Small host arrays (batch indices, mixup weights) -> device without a synchronising copy: a ring of pinned
    staging buffers, each guarded by the event of the copy that last read it."""

    def __init__(self, device, slots: int = 8):
        self.device, self.slots, self.i = device, slots, 0
        self.buf = [None] * slots
        self.done = [None] * slots

    def __call__(self, a: np.ndarray) -> torch.Tensor:
        a = np.ascontiguousarray(a)
        s = self.i = (self.i + 1) % self.slots
        if self.done[s] is not None:
            self.done[s].synchronize()
        nbytes = max(a.nbytes, 1)
        if self.buf[s] is None or self.buf[s].numel() < nbytes:
            self.buf[s] = torch.empty(max(nbytes, 1 << 16), dtype=torch.uint8).pin_memory()
        stage = self.buf[s][:a.nbytes]
        stage.numpy()[...] = a.view(np.uint8).reshape(-1)
        out = stage.to(self.device, non_blocking=True).view(torch.from_numpy(a[:0]).dtype).view(a.shape)
        self.done[s] = torch.cuda.Event()
        self.done[s].record()
        return out


class CountStore:
    """A uint8 cells x genes matrix resident in device memory.

    ``source``: path of the raw uint8 file (row-major, ``n_genes`` bytes per cell), a numpy array or a tensor.
    The file is streamed through a pinned staging buffer in ``chunk_rows``-cell pieces."""

    def __init__(self, source: Union[str, np.ndarray, torch.Tensor], n_genes: Optional[int] = None, device="cuda",
                 chunk_rows: int = 65536):
        dev = torch.device(device)
        if dev.type != "cuda":
            raise RuntimeError("vae_amd.data.CountStore: the store lives on the GPU (no CPU fallback)")
        if isinstance(source, torch.Tensor):
            if source.dtype != torch.uint8 or source.dim() != 2:
                raise TypeError("vae_amd.data.CountStore: tensor source must be 2-D uint8")
            self.dev = source.to(dev).contiguous()
        else:
            if isinstance(source, str):
                if n_genes is None:
                    raise ValueError("vae_amd.data.CountStore: n_genes is needed to read a raw uint8 file")
                host = np.memmap(source, dtype=np.uint8, mode="r")
                if host.size % n_genes:
                    raise ValueError(f"vae_amd.data.CountStore: {source} is not a whole number of {n_genes}-gene rows")
                host = host.reshape(-1, n_genes)
            else:
                host = np.asarray(source)
                if host.dtype != np.uint8 or host.ndim != 2:
                    raise TypeError("vae_amd.data.CountStore: array source must be 2-D uint8")
            n, g = host.shape
            self.dev = torch.empty(n, g, dtype=torch.uint8, device=dev)
            rows = max(1, min(chunk_rows, n))
            stage = [torch.empty(rows, g, dtype=torch.uint8).pin_memory() for _ in range(2)]
            done = [None, None]
            for c, r0 in enumerate(range(0, n, rows)):
                r1 = min(n, r0 + rows)
                s = c & 1
                if done[s] is not None:
                    done[s].synchronize()          # the copy that last used this staging buffer has finished
                stage[s][:r1 - r0].numpy()[...] = host[r0:r1]
                self.dev[r0:r1].copy_(stage[s][:r1 - r0], non_blocking=True)
                done[s] = torch.cuda.Event()
                done[s].record()
            torch.cuda.synchronize(dev)
        self.n_cells, self.n_genes = self.dev.shape
        self.device = self.dev.device
        self.upload = _Uploader(self.device)

    def gather(self, cell_rows, gene_idx: Optional[torch.Tensor] = None, dtype: torch.dtype = torch.uint8) -> torch.Tensor:
        """Rows ``cell_rows`` (host sequence / array, validated here, or a device int64 tensor) -> (B, G)."""
        if not isinstance(cell_rows, torch.Tensor):
            rows = np.asarray(cell_rows, dtype=np.int64)
            if rows.ndim != 1 or rows.size == 0:
                raise ValueError("vae_amd.data.CountStore.gather: cell_rows must be a non-empty 1-D sequence")
            if rows.min() < 0 or rows.max() >= self.n_cells:
                raise IndexError(f"vae_amd.data.CountStore.gather: row outside [0, {self.n_cells})")
            cell_rows = self.upload(rows)
        return ops.gather_counts(self.dev, cell_rows, gene_idx, dtype=dtype)


class _ArraysOnlyUnpickler(pickle.Unpickler):
    """Unpickler for the study's metadata file (data.py:41: a ``{"obs": {column: array}, "var": {...}}`` dict).
    Unpickling a file normally runs whatever callables it names; this one resolves ONLY what a dict of numpy arrays
    and plain Python containers needs, so a crafted file cannot execute code -- anything else raises."""

    _ALLOWED = {
        ("numpy.core.multiarray", "_reconstruct"), ("numpy._core.multiarray", "_reconstruct"),
        ("numpy.core.multiarray", "scalar"), ("numpy._core.multiarray", "scalar"),
        ("numpy", "ndarray"), ("numpy", "dtype"),
        ("collections", "OrderedDict"),
        ("builtins", "dict"), ("builtins", "list"), ("builtins", "tuple"), ("builtins", "set"), ("builtins", "frozenset"),
        ("builtins", "str"), ("builtins", "bytes"), ("builtins", "int"), ("builtins", "float"), ("builtins", "bool"),
        ("builtins", "complex"), ("builtins", "slice"), ("builtins", "range"),
    }

    def find_class(self, module, name):
        if (module, name) in self._ALLOWED:
            return super().find_class(module, name)
        raise pickle.UnpicklingError(
            f"vae_amd.data: the metadata file names {module}.{name}; only numpy arrays and plain containers are loaded "
            "(pass the dict itself as `metadata=`, or save it with numpy.savez)")


def _load_metadata(metadata_path, metadata):
    """The ``{"obs": .., "var": ..}`` dict itself, a ``.npz`` written by ``numpy.savez(path, **{"obs/<col>": a, "var/<col>": a})``
    (loaded with ``allow_pickle=False``), or the reference's pickle (data.py:41) through an arrays-only unpickler."""
    if metadata is not None:
        return metadata
    if str(metadata_path).endswith(".npz"):
        out: Dict[str, Dict[str, Any]] = {"obs": {}, "var": {}}
        with np.load(metadata_path, allow_pickle=False) as z:
            for k in z.files:
                table, _, col = k.partition("/")
                if table not in out or not col:
                    raise ValueError(f"vae_amd.data: unexpected key {k!r} in {metadata_path} (want obs/<column>, var/<column>)")
                out[table][col] = z[k]
        return out
    with open(metadata_path, "rb") as f:
        return _ArraysOnlyUnpickler(f).load()


def _convert_apoe(obs):
    """data.py:98-106 (only when the genotype column exists)."""
    if "ApoE_gt" not in obs:
        return
    gt = np.asarray(obs["ApoE_gt"])
    apoe = np.zeros_like(gt)
    apoe[np.where(gt == 44)[0]] = 2
    apoe[np.where((gt == 24) + (gt == 34))[0]] = 1
    apoe[np.where(np.isnan(gt))[0]] = np.nan
    obs["apoe"] = apoe


class SingleCellDataset:
    """``SingleCellDataset`` of src/data.py:18-521 over a device-resident :class:`CountStore`.

    Same constructor arguments; extra keyword-only ones: ``metadata`` (the ``{"obs": .., "var": ..}`` dict itself
    instead of a pickle path), ``store`` (share one :class:`CountStore` between the train and validation datasets),
    ``device``, ``counts_dtype`` (``torch.float32`` as the reference returns, or ``torch.uint8``: the fused step reads
    uint8 directly; mixup output is always float32), ``min_group_cells`` (250, data.py:208).

    ``dataset[batch_idx]`` returns the reference's 8-tuple with ``gene_vals``, ``cell_prop_vals``, ``cell_mask``,
    ``batch_labels`` and ``batch_mask`` as GPU tensors."""

    def __init__(self, data_path: Optional[str], metadata_path: Optional[str], cell_idx: Sequence[int],
                 cell_properties: Optional[Dict[str, Any]] = None, batch_properties: Optional[Dict[str, Any]] = None,
                 batch_size: int = 64, max_cell_prop_val: float = 999, protein_coding_only: bool = False,
                 cell_restrictions: Optional[Dict[str, Any]] = {"class": "OPC"}, max_gene_val: Optional[float] = 6.0,
                 gene_idx: Optional[Sequence[int]] = None, subject_batch_size: int = 1,
                 group_balancing: Optional[str] = "bcd", mixup: bool = True, cutmix: bool = False, training: bool = True, *,
                 metadata: Optional[Dict[str, Any]] = None, store: Optional[CountStore] = None, device="cuda",
                 counts_dtype: torch.dtype = torch.float32, min_group_cells: int = 250):
        if group_balancing is not None and group_balancing not in _GROUP_SCHEMES:
            raise NotImplementedError(f"vae_amd.data: group_balancing={group_balancing!r}; built: {sorted(_GROUP_SCHEMES)} / None")
        self.count = 0
        meta = _load_metadata(metadata_path, metadata)
        self.obs = {k: np.array(v) for k, v in meta["obs"].items()}
        var = meta["var"]
        _convert_apoe(self.obs)
        self.data_path = data_path
        self.cell_properties, self.batch_properties = cell_properties, batch_properties
        self.batch_size, self.subject_batch_size = batch_size, subject_batch_size
        self.group_balancing, self.mixup, self.cutmix, self.training = group_balancing, mixup, cutmix, training
        self.max_cell_prop_val, self.protein_coding_only, self.max_gene_val = max_cell_prop_val, protein_coding_only, max_gene_val
        self.counts_dtype = counts_dtype
        self.cell_idx = np.asarray(cell_idx)
        self._restrict_samples(cell_restrictions)
        self.n_samples = len(self.cell_idx)
        if group_balancing is not None:
            self._define_groups(min_group_cells)
        self.n_genes_original = len(var["gene_name"]) if "gene_name" in var else len(var)
        self.n_cell_properties = len(cell_properties) if cell_properties is not None else 0
        self.offset = self.n_genes_original
        self.cell_classes = np.array(['Astro', 'EN', 'Endo', 'IN', 'Immune', 'Mural', 'OPC', 'Oligo'])
        if gene_idx is None:
            self._get_gene_index(var)
        else:
            self.gene_idx = np.asarray(gene_idx)
            self.n_genes = len(self.gene_idx)
        if self.gene_idx.min() < 0 or self.gene_idx.max() >= self.n_genes_original:
            raise IndexError("vae_amd.data: gene_idx outside the store's genes")
        self.store = store if store is not None else CountStore(data_path, self.n_genes_original, device)
        if self.store.n_genes != self.n_genes_original:
            raise ValueError(f"vae_amd.data: the store has {self.store.n_genes} genes per cell, the metadata {self.n_genes_original}")
        if len(self.cell_idx) and self.cell_idx.max() >= self.store.n_cells:
            raise IndexError("vae_amd.data: cell_idx outside the store")
        dev = self.store.device
        self.device = dev
        # all genes in order -> no index array (the gather takes its 4-byte path); else the subset, on the device
        whole = self.n_genes == self.n_genes_original and np.array_equal(self.gene_idx, np.arange(self.n_genes))
        self._gene_idx_dev = None if whole else torch.from_numpy(self.gene_idx.astype(np.int32)).to(dev)
        self._get_cell_prop_vals()
        self._get_batch_prop_vals()
        self.obs = None   # as the reference drops its metadata (data.py:93)

    def __len__(self):
        return self.n_samples

    # ---- construction (host, once) ------------------------------------------------------------------
    def _restrict_samples(self, restrictions):
        """data.py:247-288: keep cells of ``cell_idx`` with include_training == 1 that meet every restriction, then
        cut every obs column down to them."""
        n_all = len(self.obs["class"]) if "class" in self.obs else len(next(iter(self.obs.values())))
        cond = np.zeros(n_all, dtype=np.uint8)
        cond[self.cell_idx] = 1
        if "include_training" in self.obs:
            cond *= self.obs["include_training"] == 1
        for k, v in (restrictions or {}).items():
            if k in ("positive_prs", "negative_prs"):
                raise NotImplementedError("vae_amd.data: the polygenic-score restrictions (data.py:272-278) are not built")
            if isinstance(v, list):
                cond *= np.sum(np.stack([self.obs[k] == v1 for v1 in v]), axis=0).astype(np.uint8)
            else:
                cond *= self.obs[k] == v
        self.cell_idx = np.where(cond)[0]
        for k in self.obs:
            self.obs[k] = self.obs[k][self.cell_idx]

    def _define_groups(self, min_cells):
        """data.py:109-215, two-column schemes: (staging value) x (Dementia 0 / 1), key 10 b + d; 999 = unknown."""
        col, values = _GROUP_SCHEMES[self.group_balancing]
        dem, stage = self.obs["Dementia"], self.obs[col]
        groups = {}
        for b in values:
            for d in range(2):
                key = (b - 1) * 10 + d if self.group_balancing == "cd" else b * 10 + d
                groups[key] = np.where((dem == d) * (stage == b))[0].tolist()
        groups[999] = np.where((dem < 0) + (stage < 0))[0].tolist()
        self.group_idx = {k: v for k, v in groups.items() if len(v) >= min_cells}
        if not self.group_idx:
            raise ValueError("vae_amd.data: no group reaches min_group_cells")
        self.cell_per_group = self.batch_size // len(self.group_idx)

    def _get_gene_index(self, var):
        """data.py:299-325: every gene, or (protein_coding_only) those expressed, off the sex chromosomes and not
        MALAT1 / LINC00486."""
        if self.protein_coding_only:
            cond = np.array([g not in ["MALAT1", "LINC00486"] for g in var["gene_name"]])
            cond = cond * (np.asarray(var["percent_cells"]) > 0.0) * (np.asarray(var["gene_chrom"]) != "X") * \
                (np.asarray(var["gene_chrom"]) != "Y")
            self.gene_idx = np.where(cond)[0]
        else:
            self.gene_idx = np.arange(self.n_genes_original)
        self.n_genes = len(self.gene_idx)
        if "gene_name" in var:
            self.gene_names = np.asarray(var["gene_name"])[self.gene_idx]

    def _get_cell_prop_vals(self):
        """data.py:350-390, column-wise instead of cell by cell; the tables then live on the device."""
        if self.n_cell_properties == 0:
            self.labels = self.mask = None
            self.subjects = np.asarray(self.obs["SubID"]) if "SubID" in self.obs else None
            return
        n, P = self.n_samples, self.n_cell_properties
        labels = np.zeros((n, P), dtype=np.float32)
        mask = np.ones((n, P), dtype=np.float32)
        for n1, (k, prop) in enumerate(self.cell_properties.items()):
            vals = self.obs[k]
            if not prop["discrete"]:
                v = vals.astype(np.float64)
                bad = (v > self.max_cell_prop_val) | (v < -self.max_cell_prop_val) | np.isnan(v)
                with np.errstate(invalid="ignore"):
                    labels[:, n1] = np.where(bad, -100.0, (v - prop["mean"]) / prop["std"])
                mask[bad, n1] = 0.0
            else:
                choices = np.array(prop["values"])
                hit = vals[:, None] == choices[None, :]
                found = hit.any(axis=1)
                labels[:, n1] = np.where(found, hit.argmax(axis=1), -100)
                mask[~found, n1] = 0.0
        self.labels, self.mask = labels, mask
        self.subjects = np.asarray(self.obs["SubID"]) if "SubID" in self.obs else None
        self.unique_subjects = np.unique(self.subjects) if self.subjects is not None else None
        if "class" in self.obs:
            hit = self.obs["class"][:, None] == self.cell_classes[None, :]
            self.cell_class = hit.argmax(axis=1).astype(np.uint8)
        if (self.mixup or self.cutmix) and self.training:
            for n1, (k, prop) in enumerate(self.cell_properties.items()):
                if prop["discrete"] and (mask[:, n1] == 0).any():
                    # alpha * (-100) + (1 - alpha) * class is no class index: cross_entropy then reads outside its
                    # logits (a device fault); the reference has the same hazard (data.py:518 "does not work with ...")
                    raise ValueError(f"vae_amd.data: mixup with unknown values in the discrete property {k!r}: "
                                     "their label -100 would be blended into class indices")
        self._labels_dev = torch.from_numpy(labels).to(self.device)
        self._mask_dev = torch.from_numpy(mask).to(self.device)

    def _get_batch_prop_vals(self):
        """data.py:327-348."""
        if self.batch_properties is None:
            self.batch_labels = self.batch_mask = None
            return
        n, P = self.n_samples, len(self.batch_properties)
        labels = np.zeros((n, P), dtype=np.int64)
        mask = np.ones((n, P), dtype=np.int64)
        for n1, (k, prop) in enumerate(self.batch_properties.items()):
            hit = self.obs[k][:, None] == np.array(prop["values"])[None, :]
            found = hit.any(axis=1)
            labels[:, n1] = np.where(found, hit.argmax(axis=1), -100)
            mask[~found, n1] = 0
        self.batch_labels, self.batch_mask = labels, mask
        self._batch_labels_dev = torch.from_numpy(labels).to(self.device)
        self._batch_mask_dev = torch.from_numpy(mask).to(self.device)

    def library_size_stats(self, N: int = 200_000):
        """data.py:233-244: mean / variance of log1p(row sum over the gene subset) of the first N cells, on the device."""
        idx = self.cell_idx[:N] if len(self.cell_idx) > N else self.cell_idx
        logs = []
        for r0 in range(0, len(idx), 16384):
            x = self.store.gather(idx[r0:r0 + 16384], self._gene_idx_dev, torch.float32)
            logs.append(torch.log1p(x.sum(dim=1)).double())
        logs = torch.cat(logs)
        return float(logs.mean()), float(logs.var(unbiased=False))

    # ---- one batch -----------------------------------------------------------------------------------
    def _get_balanced_batch(self, old_batch_idx):
        """data.py:480-495: the reference's draws (``np.random.choice(v, size, replace=False)`` per group, dict order)."""
        batch_idx: List[int] = []
        group_idx: List[int] = []
        for k, v in self.group_idx.items():
            batch_idx += np.random.choice(v, size=self.cell_per_group, replace=False).tolist()
            group_idx += [k] * self.cell_per_group
        return batch_idx, group_idx

    def _mixup_plan(self, N):
        """The random part of data.py:457-472 with the reference's global-state draws: two gamma vectors, then one
        bounded integer per cell (``np.random.choice(list)`` is one ``randint(0, len)``; N of them in one call consume
        the generator identically).  Returns (alpha float64 (N), partner int64 (N))."""
        g1 = np.random.gamma(0.25, size=N)
        g2 = np.random.gamma(0.25, size=N)
        alpha = g1 / (g1 + g2)
        if self.group_balancing is not None:
            cpg = self.cell_per_group
            partner = np.random.randint(0, cpg, size=N) + (np.arange(N) // cpg) * cpg
        else:
            partner = np.random.randint(0, N, size=N)
        return alpha, partner

    def _mixup(self, gene_vals, cell_vals, cell_mask):
        """data.py:457-477 on the device; bit-identical to the numpy arithmetic (``vae_mixup_counts``)."""
        N = gene_vals.shape[0]
        alpha, partner = self._mixup_plan(N)
        if partner.max() >= N:
            raise ValueError("vae_amd.data: batch smaller than groups x cell_per_group, mixup partner outside the batch")
        a = self.store.upload(alpha)
        p = self.store.upload(partner.astype(np.int32))
        gene_vals = ops.mixup_counts(gene_vals, p, a)
        if cell_vals is not None:
            cell_vals = ops.mixup_counts(cell_vals, p, a)
            cell_mask = ops.mixup_counts(cell_mask, p, a)
        return gene_vals, cell_vals, cell_mask

    def _cutmix(self, gene_vals, cell_vals, cell_mask, keep_u8=False):
        """data.py:431-454: the reference's draws replayed on the host in its order (two gamma vectors, then per cell one
        partner and one ``np.random.choice(n_genes, int(alpha n_genes), replace=False)``), the arrays on the device:
        genes of the subset are COPIED from the partner (``vae_cutmix_counts``; counts stay integers), the label / mask
        tables are blended with alpha as in mixup."""
        N, G = gene_vals.shape
        g1 = np.random.gamma(0.25, size=N)
        g2 = np.random.gamma(0.25, size=N)
        alpha = g1 / (g1 + g2)
        partner = np.zeros(N, dtype=np.int64)
        take = np.zeros((N, G), dtype=np.uint8)
        cpg = self.cell_per_group if self.group_balancing is not None else None
        for i in range(N):
            partner[i] = np.random.randint(0, cpg) + (i // cpg) * cpg if cpg is not None else np.random.randint(0, N)
            take[i, np.random.choice(self.n_genes, int(alpha[i] * self.n_genes), replace=False)] = 1
        if partner.max() >= N:
            raise ValueError("vae_amd.data: batch smaller than groups x cell_per_group, cutmix partner outside the batch")
        p = self.store.upload(partner.astype(np.int32))
        gene_vals = ops.cutmix_counts(gene_vals, p, torch.from_numpy(take).to(self.device),
                                      out_dtype=gene_vals.dtype if (keep_u8 or self.counts_dtype != torch.float32) else torch.float32)
        if cell_vals is not None:
            a = self.store.upload(alpha)
            cell_vals = ops.mixup_counts(cell_vals, p, a)
            cell_mask = ops.mixup_counts(cell_mask, p, a)
        return gene_vals, cell_vals, cell_mask

    def __getitem__(self, batch_idx):
        if isinstance(batch_idx, (int, np.integer)):
            batch_idx = [int(batch_idx)]
        if self.training and self.group_balancing is not None and self.count > -1:
            batch_idx, group_idx = self._get_balanced_batch(batch_idx)
        else:
            group_idx = None
        self.count += 1
        rows = np.asarray(batch_idx, dtype=np.int64)
        if rows.size == 0 or rows.min() < 0 or rows.max() >= self.n_samples:
            raise IndexError(f"vae_amd.data: batch index outside [0, {self.n_samples})")
        cell_idx = self.cell_idx[rows]
        mix = self.mixup and self.training
        cut = self.cutmix and self.training                    # data.py:513-519: cutmix first, then mixup (both may be on)
        gene_vals = self.store.gather(cell_idx, self._gene_idx_dev, torch.uint8 if (mix or cut) else self.counts_dtype)
        cell_prop_vals = cell_mask = batch_labels = batch_mask = None
        if self.labels is not None or self.batch_properties is not None:
            rows_dev = self.store.upload(rows)
        if self.labels is not None:
            cell_prop_vals, cell_mask = self._labels_dev[rows_dev], self._mask_dev[rows_dev]
        if self.batch_properties is not None:
            batch_labels, batch_mask = self._batch_labels_dev[rows_dev], self._batch_mask_dev[rows_dev]
        subject = self.subjects[rows] if self.subjects is not None else None
        if cut:
            gene_vals, cell_prop_vals, cell_mask = self._cutmix(gene_vals, cell_prop_vals, cell_mask, keep_u8=mix)
        if mix:
            gene_vals, cell_prop_vals, cell_mask = self._mixup(gene_vals, cell_prop_vals, cell_mask)
        return (gene_vals, cell_prop_vals, cell_mask, batch_labels, batch_mask, cell_idx, group_idx, subject)


class _Loader:
    """What ``DataLoader(dataset, batch_size=None, sampler=BatchSampler(...))`` iterates (data.py:700-729), without
    worker processes: a batch is a device gather, there is nothing for workers to do."""

    def __init__(self, dataset, sampler):
        self.dataset, self.sampler = dataset, sampler

    def __len__(self):
        return len(self.sampler)

    def __iter__(self):
        for batch_idx in self.sampler:
            yield self.dataset[batch_idx]


class DataModule:
    """``DataModule`` of src/data.py:524-729 (a plain class: Lightning is not a dependency)."""

    def __init__(self, data_path: Optional[str], metadata_path: Optional[str], train_idx: Sequence[int], test_idx: Sequence[int],
                 batch_size: int = 32, num_workers: int = 16, cell_properties: Optional[Dict[str, Any]] = None,
                 batch_properties: Optional[Dict[str, Any]] = None, protein_coding_only: bool = False,
                 cell_restrictions: Optional[Dict[str, Any]] = None, group_balancing: Optional[str] = "bcd",
                 mixup: bool = False, *, metadata: Optional[Dict[str, Any]] = None, store: Optional[CountStore] = None,
                 device="cuda", counts_dtype: torch.dtype = torch.float32, min_group_cells: int = 250):
        self.data_path, self.metadata_path = data_path, metadata_path
        self.train_idx, self.test_idx = train_idx, test_idx
        self.batch_size, self.num_workers = batch_size, num_workers     # num_workers: accepted, unused
        self.cell_properties, self.batch_properties = cell_properties, batch_properties
        self.protein_coding_only, self.cell_restrictions = protein_coding_only, cell_restrictions
        self.group_balancing, self.mixup = group_balancing, mixup
        self._metadata = _load_metadata(metadata_path, metadata)
        self._store, self._device, self._counts_dtype, self._min_group_cells = store, device, counts_dtype, min_group_cells
        self._get_cell_prop_info()
        self._get_batch_prop_info()

    def _obs(self):
        obs = {k: np.array(v) for k, v in self._metadata["obs"].items()}
        _convert_apoe(obs)
        return obs

    def _get_batch_prop_info(self):
        """data.py:564-591: value lists of batch properties that came without one."""
        if self.batch_properties is None:
            return
        obs = self._obs()
        for k, prop in self.batch_properties.items():
            if prop["values"] is None:
                uniq = np.unique(obs[k])
                keep = [n for n, u in enumerate(uniq) if (isinstance(u, str) or (u >= 0 and u < 999))]
                prop["values"] = uniq[keep]

    def _get_cell_prop_info(self, max_cell_prop_val=999):
        """data.py:594-659: mean / std of continuous properties, values / freq of discrete ones."""
        self.n_cell_properties = len(self.cell_properties) if self.cell_properties is not None else 0
        var = self._metadata["var"]
        self.n_genes = len(var["gene_name"]) if "gene_name" in var else len(var)
        if self.n_cell_properties == 0:
            self.cell_prop_dist = None
            return
        obs = self._obs()
        for k, prop in self.cell_properties.items():
            vals = obs[k]
            if "freq" in prop or "mean" in prop:
                continue
            if not prop["discrete"]:
                ok = (vals >= 0) & (vals < max_cell_prop_val)
                prop["mean"], prop["std"] = np.mean(vals[ok]), np.std(vals[ok])
            else:
                uniq, counts = np.unique(vals, return_counts=True)
                if prop["values"] is None:
                    keep = [n for n, u in enumerate(uniq) if (isinstance(u, str) or (u >= 0 and u < max_cell_prop_val))]
                    prop["values"] = uniq[keep]
                else:
                    keep = [n for n, u in enumerate(uniq) if u in prop["values"]]
                prop["freq"] = counts[keep] / np.mean(counts[keep])

    def setup(self, stage=None):
        """data.py:665-696: the two datasets share one device-resident store."""
        common = dict(cell_properties=self.cell_properties, batch_properties=self.batch_properties,
                      batch_size=self.batch_size, protein_coding_only=self.protein_coding_only,
                      cell_restrictions=self.cell_restrictions, metadata=self._metadata, device=self._device,
                      counts_dtype=self._counts_dtype, min_group_cells=self._min_group_cells)
        self.train_dataset = SingleCellDataset(self.data_path, self.metadata_path, self.train_idx, store=self._store,
                                               group_balancing=self.group_balancing, mixup=self.mixup, training=True, **common)
        self.val_dataset = SingleCellDataset(self.data_path, self.metadata_path, self.test_idx, store=self.train_dataset.store,
                                             gene_idx=self.train_dataset.gene_idx, group_balancing=None, mixup=False,
                                             training=False, **common)
        self.n_genes = self.train_dataset.n_genes

    def train_dataloader(self, generator: Optional[torch.Generator] = None):
        """``generator``: the sampler's own source of randomness (default: torch's global generator, as the
        reference).  Data-parallel ranks pass one seeded per rank, so that every rank draws its own cells --
        what Lightning's DDP does with a per-rank sampler shard (train.py:57)."""
        return _Loader(self.train_dataset, BatchSampler(RandomSampler(self.train_dataset, generator=generator),
                                                        batch_size=self.train_dataset.batch_size, drop_last=True))

    def val_dataloader(self):
        return _Loader(self.val_dataset, BatchSampler(SequentialSampler(self.val_dataset), batch_size=self.val_dataset.batch_size,
                                                      drop_last=False))
