"""Deterministic, platform-independent inputs for the golden fixtures.

Shared by ``make_goldens.py`` (which feeds them to the REFERENCE in the build container)
and by the tests (which feed the same inputs to the oracle and to the HIP path).  All
values come from integer hashing + exact IEEE arithmetic (no libm, no torch RNG), so the
two sides regenerate bit-identical tensors and large inputs need not be stored.
"""
from __future__ import annotations

import math
from collections import OrderedDict

import numpy as np
import torch

_MASK = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix(idx: np.ndarray, stream: int) -> np.ndarray:
    """splitmix64 of (idx, stream) -> uint64."""
    with np.errstate(over="ignore"):
        z = idx.astype(np.uint64) + np.uint64((stream * 0x9E3779B97F4A7C15 + 0x1234567) & 0xFFFFFFFFFFFFFFFF)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return z


def _uniform(n: int, stream: int) -> np.ndarray:
    """n doubles in [0,1) with 32 random bits each (exactly representable)."""
    return (_splitmix(np.arange(n, dtype=np.uint64), stream) >> np.uint64(32)).astype(np.float64) / 4294967296.0


def formula_tensor(shape, stream: int, kind: str = "uniform", scale: float = 1.0) -> torch.Tensor:
    """float64 tensor: 'uniform' in [-scale, scale); 'normal' ~ Irwin-Hall(12)-6 times scale."""
    n = int(np.prod(shape))
    if kind == "uniform":
        v = (_uniform(n, stream) * 2.0 - 1.0) * scale
    elif kind == "normal":
        v = np.zeros(n)
        for j in range(12):
            v += _uniform(n, stream * 16 + j + 1000)
        v = (v - 6.0) * scale
    else:
        raise ValueError(kind)
    return torch.from_numpy(v.reshape(shape))


def formula_counts(B: int, G: int, stream: int, zero_frac: float = 0.35) -> torch.Tensor:
    """Integer-valued count matrix (float64): heavy-tailed, ~zero_frac exact zeros plus the
    zeros of the floor; every row has at least one positive entry."""
    n = B * G
    u1 = _uniform(n, stream)
    u2 = _uniform(n, stream + 1)
    k = (_splitmix(np.arange(G, dtype=np.uint64), stream + 2) % np.uint64(7)).astype(np.int64) - 4  # -4..2
    gene_scale = np.ldexp(1.0, k)                       # exact powers of two
    t = u1 / (1.0 - u1)                                 # IEEE division, deterministic
    x = np.floor(np.minimum(t.reshape(B, G) * gene_scale[None, :], 3000.0))
    x = np.where(u2.reshape(B, G) < zero_frac, 0.0, x)
    x[:, 0] = np.maximum(x[:, 0], 1.0)
    return torch.from_numpy(x)


def _lin(p, name, n_out, n_in, stream):
    b = 1.0 / math.sqrt(n_in)
    p[f"{name}.weight"] = formula_tensor((n_out, n_in), stream, "uniform", b)
    p[f"{name}.bias"] = formula_tensor((n_out,), stream + 1, "uniform", b)
    return stream + 2


def _ln(p, name, n, stream):
    p[f"{name}.weight"] = 1.0 + formula_tensor((n,), stream, "uniform", 0.2)
    p[f"{name}.bias"] = formula_tensor((n,), stream + 1, "uniform", 0.1)
    return stream + 2


def formula_params(G, H, Hd, Hl, L, n_layers, cell_properties, stream0=100):
    """float64 parameters under the reference VAE's state_dict keys."""
    p = OrderedDict()
    s = stream0
    p["px_r"] = formula_tensor((G,), s, "normal", 1.0); s += 1
    for n in range(n_layers):
        s = _lin(p, f"z_encoder.encoder.fc_layers.{4 * n}", H, G if n == 0 else H, s)
        s = _ln(p, f"z_encoder.encoder.fc_layers.{4 * n + 2}", H, s)
    s = _lin(p, "z_encoder.mean_encoder", L, H, s)
    s = _lin(p, "z_encoder.var_encoder", 1, H, s)
    for k, prop in cell_properties.items():
        n_t = len(prop["values"]) if prop["discrete"] else 1
        s = _lin(p, f"cell_decoder.cell_mlp.{k}", n_t, L, s)
    s = _lin(p, "l_encoder.encoder.fc_layers.0", Hl, G, s)
    s = _ln(p, "l_encoder.encoder.fc_layers.2", Hl, s)
    s = _lin(p, "l_encoder.mean_encoder", 1, Hl, s)
    s = _lin(p, "l_encoder.var_encoder", 1, Hl, s)
    for n in range(n_layers):
        s = _lin(p, f"decoder.px_decoder.fc_layers.{4 * n}", Hd, L if n == 0 else Hd, s)
        s = _ln(p, f"decoder.px_decoder.fc_layers.{4 * n + 2}", Hd, s)
    s = _lin(p, "decoder.px_scale_decoder.0", G, Hd, s)
    s = _lin(p, "decoder.px_r_decoder", G, Hd, s)
    s = _lin(p, "decoder.px_dropout_decoder", G, Hd, s)
    return p


_ONE_PROP = OrderedDict(A={"discrete": True, "values": [0, 1], "stop_grad": False})
_HEADS = OrderedDict(
    A={"discrete": True, "values": [0, 1], "stop_grad": False},
    Bc={"discrete": False, "values": [-1], "stop_grad": True},
    C={"discrete": True, "values": [0, 1, 2], "stop_grad": False},
    Dc={"discrete": False, "values": [-1], "stop_grad": False},
    E={"discrete": True, "values": [0, 1], "stop_grad": True},
)


def _case(name, likelihood, n_layers, heads, **kw):
    c = dict(name=name, likelihood=likelihood, n_layers=n_layers, B=8, G=64, H=16, Hd=16, Hl=8, L=4,
             cell_properties=_HEADS if heads else _ONE_PROP, use_cell_loss=heads,
             grad_reverse={"C": 0.5, "E": 0.2} if heads else {}, kl_weight=1.0, gene_loss_coeff=5e-3,
             sample_stride=1, stream=7)
    c.update(kw)
    return c


E2E_CASES = [
    _case("nb_l1", "nb", 1, False),
    _case("zinb_l1", "zinb", 1, False, stream=11),
    _case("zinb_l2", "zinb", 2, False, stream=13),
    _case("zinb_l2_heads", "zinb", 2, True, kl_weight=0.7, stream=17),
    _case("nb_l1_heads", "nb", 1, True, gene_loss_coeff=5e-4, stream=19),
    # ragged sizes: nothing a multiple of a tile / wave / vector width
    _case("zinb_ragged", "zinb", 1, False, B=5, G=77, H=12, Hd=10, Hl=6, L=3, stream=23),
]
WIDE_CASE = _case("zinb_wide20k", "zinb", 1, False, B=4, G=20000, H=8, Hd=8, Hl=4, L=3,
                  sample_stride=97, stream=29)

INIT_CASE = dict(G=48, H=12, Hd=10, Hl=6, L=3, n_layers=2, cell_properties=_HEADS, grad_reverse={"C": 0.5, "E": 0.2})

# x, mu, theta, pi grid of SURVEY.md §8c(i) (softplus-threshold edges at |pi| ~ 20)
LIKELIHOOD_GRID = (
    np.array([0, 1, 2, 5, 50, 255, 1e3, 5e4]),
    np.array([1e-8, 1e-4, 1e-2, 0.5, 3.0, 40.0, 1e3, 1e4]),
    np.array([1e-3, 5e-2, 0.7, 1.0, 4.5, 60.0, 1e3, 1e4]),
    np.array([-30, -20.5, -5, 0, 5, 19.9, 20.1, 30.0]),
)


def build_case_inputs(case):
    """-> (params64, x64, eps_z64, eps_l64, targets, mask); also fills case['l_mean'/'l_var']."""
    B, G, L = case["B"], case["G"], case["L"]
    s = case["stream"] * 1000
    params = formula_params(G, case["H"], case["Hd"], case["Hl"], L, case["n_layers"], case["cell_properties"], s)
    x = formula_counts(B, G, s + 500)
    eps_z = formula_tensor((B, L), s + 600, "normal")
    eps_l = formula_tensor((B, 1), s + 601, "normal")
    lib = np.log1p(x.numpy().astype(np.float32).sum(axis=1))
    case["l_mean"], case["l_var"] = float(np.mean(lib)), float(max(np.var(lib), 0.05))
    P = len(case["cell_properties"])
    targets = torch.zeros(B, P, dtype=torch.float32)
    mask = torch.ones(B, P, dtype=torch.float32)
    u = _uniform(B * P, s + 700).reshape(B, P)
    for n, (k, prop) in enumerate(case["cell_properties"].items()):
        if prop["discrete"]:
            targets[:, n] = torch.from_numpy(np.floor(u[:, n] * len(prop["values"]))).float()
        else:
            targets[:, n] = torch.from_numpy(u[:, n] * 4 - 2).float()
    mask[0, :] = 0.0
    if P > 1:
        mask[1, 1] = 0.0
        targets[0, 0] = -100.0  # ignore_index path (task.py:132), masked out as the loader would
    return params, x, eps_z, eps_l, targets, mask


# ---- row f4: batch covariates (modules.py:121-130) and GroupedEncoder (modules.py:134-188) ----------------------
_F4_PROPS = OrderedDict(
    A={"discrete": True, "values": [0, 1], "stop_grad": False},
    Dc={"discrete": False, "values": [-1], "stop_grad": False},
    C={"discrete": True, "values": [0, 1, 2], "stop_grad": True},
)
_F4_BATCH = OrderedDict(Brain_bank={"values": ["A", "B", "C"]}, Sex={"values": ["M", "F"]})
F4_CASES = [
    dict(name="batchcov", G=64, H=16, Hd=12, Hl=8, L=4, L_cell=4, n_layers=2, B=12, likelihood="zinb", seed=101,
         cell_properties=_F4_PROPS, batch_properties=_F4_BATCH, grouped=False, grad_reverse={"C": 0.3}, gene_loss_coeff=5e-3,
         stream=41),
    # grouped: the decoder and the cell heads see the concatenation of one latent per property (L * 3)
    dict(name="grouped", G=64, H=16, Hd=12, Hl=8, L=3, L_cell=9, n_layers=1, B=10, likelihood="zinb", seed=102,
         cell_properties=_F4_PROPS, batch_properties=None, grouped=True, grad_reverse={}, gene_loss_coeff=5e-3, stream=43),
]


def build_f4_inputs(case):
    """-> (x64, [eps_z64 per z-encoder], eps_l64, batch_labels i64|None, batch_mask i64|None, targets f32, mask f32);
    fills case['l_mean'/'l_var'].  batch labels contain -1 (unknown: the reference clamps them to 0 IN PLACE and the mask
    removes them) and fully masked rows."""
    B, G, L = case["B"], case["G"], case["L"]
    s = case["stream"] * 1000
    x = formula_counts(B, G, s + 500)
    n_enc = len(case["cell_properties"]) if case["grouped"] else 1
    eps_z = [formula_tensor((B, L), s + 600 + 10 * j, "normal") for j in range(n_enc)]
    eps_l = formula_tensor((B, 1), s + 601, "normal")
    lib = np.log1p(x.numpy().astype(np.float32).sum(axis=1))
    case["l_mean"], case["l_var"] = float(np.mean(lib)), float(max(np.var(lib), 0.05))
    batch_labels = batch_mask = None
    if case["batch_properties"]:
        nb = len(case["batch_properties"])
        u = _uniform(B * nb, s + 800).reshape(B, nb)
        sizes = np.array([len(v["values"]) for v in case["batch_properties"].values()])
        lab = np.floor(u * sizes[None, :]).astype(np.int64)
        m = np.ones((B, nb), dtype=np.int64)
        lab[1, 0], m[1, 0] = -1, 0          # unknown label: clamped to 0 by the reference, masked out
        m[2, :] = 0                          # a cell with no covariate at all
        batch_labels, batch_mask = torch.from_numpy(lab), torch.from_numpy(m)
    P = len(case["cell_properties"])
    targets = torch.zeros(B, P, dtype=torch.float32)
    mask = torch.ones(B, P, dtype=torch.float32)
    u = _uniform(B * P, s + 700).reshape(B, P)
    for n, (k, prop) in enumerate(case["cell_properties"].items()):
        if prop["discrete"]:
            targets[:, n] = torch.from_numpy(np.floor(u[:, n] * len(prop["values"]))).float()
        else:
            targets[:, n] = torch.from_numpy(u[:, n] * 4 - 2).float()
    mask[0, :] = 0.0
    targets[0, 0] = -100.0
    return x, eps_z, eps_l, batch_labels, batch_mask, targets, mask
