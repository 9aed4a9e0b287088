"""Host logic of vae_amd.data (rows f2 / f4) against oracle/data_oracle.py: sample restriction, groups, label tables,
the balanced sampler and the mixup draws consume np.random exactly as the reference's code does (data.py)."""
import copy

import numpy as np
import pytest

from helpers import HostOnlyStore, synthetic_study
from oracle import data_oracle as DO
from vae_amd.data import DataModule, SingleCellDataset

CELL_PROPS = {"Age": {"discrete": False, "values": None}, "Sex": {"discrete": True, "values": ["M", "F"]},
              "Dementia": {"discrete": True, "values": [0, 1]}}
BATCH_PROPS = {"Brain_bank": {"values": ["A", "B"]}}


def _dataset(store, meta, cell_props, **kw):
    args = dict(cell_properties=cell_props, batch_properties=copy.deepcopy(BATCH_PROPS), batch_size=64,
                cell_restrictions={"class": "OPC"}, group_balancing="bcd", mixup=True, training=True, metadata=meta,
                store=HostOnlyStore(store), min_group_cells=40)
    args.update(kw)
    return SingleCellDataset(None, None, np.arange(0, store.shape[0], 2), **args)


def _filled_props(meta):
    obs = {k: np.array(v) for k, v in meta["obs"].items()}
    return DO.cell_prop_info(obs, copy.deepcopy(CELL_PROPS))


def test_restriction_groups_and_tables_match_the_oracle():
    store, meta = synthetic_study()
    props = _filled_props(meta)
    ds = _dataset(store, meta, props, mixup=False)
    with pytest.raises(ValueError):       # mixup would blend the N/A label -100 of unknown sexes into class indices
        _dataset(store, meta, props, mixup=True)
    obs = {k: np.array(v) for k, v in meta["obs"].items()}
    keep = DO.restrict_samples(obs, np.arange(0, store.shape[0], 2), {"class": "OPC"})
    assert np.array_equal(ds.cell_idx, keep) and len(ds) == len(keep) and 500 < len(keep) < 1500
    sub = {k: v[keep] for k, v in obs.items()}
    groups, cpg = DO.define_groups(sub, "bcd", 64, min_cells=40)
    assert ds.group_idx == groups and ds.cell_per_group == cpg and 999 in groups and len(groups) >= 5
    labels, mask = DO.cell_prop_arrays(sub, props)
    assert np.array_equal(ds.labels, labels) and np.array_equal(ds.mask, mask)
    assert (mask == 0).any() and (labels == -100).any()
    bl, bm = DO.batch_prop_arrays(sub, BATCH_PROPS)
    assert np.array_equal(ds.batch_labels, bl) and np.array_equal(ds.batch_mask, bm) and (bm == 0).any()
    assert np.array_equal(ds.subjects, sub["SubID"])
    # a list-valued restriction and the gene filter
    ds2 = _dataset(store, meta, props, cell_restrictions={"class": ["OPC", "EN"]}, group_balancing=None, protein_coding_only=True,
                   mixup=False)
    assert np.array_equal(ds2.cell_idx, DO.restrict_samples(obs, np.arange(0, store.shape[0], 2), {"class": ["OPC", "EN"]}))
    var = meta["var"]
    want = np.where((var["gene_name"] != "MALAT1") & (var["percent_cells"] > 0) & (var["gene_chrom"] != "X") &
                    (var["gene_chrom"] != "Y"))[0]
    assert np.array_equal(ds2.gene_idx, want) and ds2.n_genes == len(want) < store.shape[1]


def test_balanced_batches_and_mixup_draws_follow_the_reference_stream():
    store, meta = synthetic_study()
    props = {k: v for k, v in _filled_props(meta).items() if k != "Dementia"}    # unknown Dementia stays: group 999
    ds = _dataset(store, meta, props, cell_restrictions={"class": "OPC", "Sex": ["M", "F"]})
    N = ds.cell_per_group * len(ds.group_idx)
    for seed in (0, 7):
        np.random.seed(seed)
        want_idx, want_groups = DO.balanced_batch(ds.group_idx, ds.cell_per_group)
        g = np.zeros((N, 3), np.float32)
        _, _, _, want_alpha, want_partner = DO.mixup(g, g[:, :2], g[:, :2], ds.cell_per_group)
        after_oracle = np.random.rand()
        np.random.seed(seed)
        got_idx, got_groups = ds._get_balanced_batch(None)
        alpha, partner = ds._mixup_plan(N)
        assert got_idx == want_idx and got_groups == want_groups and len(set(got_idx)) == N
        assert np.array_equal(alpha, want_alpha) and np.array_equal(partner, want_partner)
        assert np.random.rand() == after_oracle          # the generator was consumed identically
        assert np.array_equal(partner // ds.cell_per_group, np.arange(N) // ds.cell_per_group)
    # without groups the partner is any cell of the batch
    ds.group_balancing = None
    np.random.seed(3)
    _, _, _, want_alpha, want_partner = DO.mixup(g, g[:, :2], g[:, :2], None)
    np.random.seed(3)
    alpha, partner = ds._mixup_plan(N)
    assert np.array_equal(alpha, want_alpha) and np.array_equal(partner, want_partner)


def test_datamodule_fills_the_property_tables_like_the_oracle():
    store, meta = synthetic_study()
    props = copy.deepcopy(CELL_PROPS)
    props["Sex"]["values"] = None
    bprops = {"Brain_bank": {"values": None}}
    dm = DataModule(None, None, np.arange(2000), np.arange(2000, 3000), batch_size=64, cell_properties=props,
                    batch_properties=bprops, cell_restrictions={"class": "OPC"}, group_balancing="bcd", metadata=meta,
                    store=HostOnlyStore(store), min_group_cells=40)
    obs = {k: np.array(v) for k, v in meta["obs"].items()}
    raw = copy.deepcopy(CELL_PROPS)
    raw["Sex"]["values"] = None
    want = DO.cell_prop_info(obs, raw)
    for k in want:
        for f in ("mean", "std", "freq", "values"):
            if f in want[k] and want[k][f] is not None:
                assert np.array_equal(np.asarray(dm.cell_properties[k][f]), np.asarray(want[k][f])), (k, f)
    assert list(dm.cell_properties["Sex"]["values"]) == ["F", "M", "U"]
    assert list(bprops["Brain_bank"]["values"]) == ["A", "B", "C"]
    assert dm.n_genes == store.shape[1]


def test_unbuilt_variants_raise():
    store, meta = synthetic_study(n_cells=400)
    # (cutmix is built since round 2: tests/test_data_gpu.py; these need no cell properties)
    with pytest.raises(NotImplementedError):
        _dataset(store, meta, None, group_balancing="bd_special")
    with pytest.raises(NotImplementedError):
        _dataset(store, meta, None, cell_restrictions={"positive_prs": True}, group_balancing=None)
    with pytest.raises(IndexError):
        _dataset(store, meta, None, gene_idx=[0, 5000], group_balancing=None)


def test_metadata_loader_executes_nothing_from_the_file(tmp_path):
    """the metadata file (data.py:41) was read with pickle.load, which runs whatever the file names.  The
    loader now resolves numpy arrays and plain containers only, and also reads an .npz (allow_pickle=False)."""
    import pickle
    from vae_amd.data import _load_metadata
    _, meta = synthetic_study(n_cells=50, n_genes=20)
    f = tmp_path / "metadata.pkl"
    f.write_bytes(pickle.dumps(meta))
    got = _load_metadata(str(f), None)
    assert set(got) == {"obs", "var"}
    for table in ("obs", "var"):
        for k, v in meta[table].items():
            assert np.array_equal(np.asarray(got[table][k]), np.asarray(v), equal_nan=np.asarray(v).dtype.kind == "f"), k

    class Boom:
        def __reduce__(self):
            import os
            return (os.system, ("echo pwned > /dev/null",))

    bad = tmp_path / "bad.pkl"
    bad.write_bytes(pickle.dumps({"obs": {"x": Boom()}, "var": {}}))
    with pytest.raises(pickle.UnpicklingError, match="only numpy arrays"):
        _load_metadata(str(bad), None)
    z = tmp_path / "metadata.npz"
    np.savez(z, **{f"{t}/{k}": np.asarray(v) for t in ("obs", "var") for k, v in meta[t].items()})
    got = _load_metadata(str(z), None)
    assert np.array_equal(got["obs"]["SubID"], meta["obs"]["SubID"]) and np.array_equal(got["var"]["gene_name"], meta["var"]["gene_name"])
    assert _load_metadata(None, meta) is meta
