"""Flat parameter / gradient arena for the VAE hot path.

All parameters of the model live in ONE contiguous fp32 buffer in HBM, ordered so that
  * the first-layer weights of the z- and library-encoders are adjacent: one (H+Hl, G) matrix,
    so the counts are streamed through ONE genes->hidden GEMM;
  * the px_scale / px_dropout head weights are adjacent: one (2G, Hd) matrix -> one
    hidden->genes GEMM producing both logit arrays;
  * every small tensor whose gradient is accumulated with float atomics (biases, LayerNorm
    affine, px_r, cell heads) sits in one "small" region that is zeroed with a single fill
    per step; the big weight gradients are overwritten by the wgrad GEMMs and never zeroed.
The gradient arena mirrors the layout, which makes the data-parallel all-reduce a handful
of large contiguous buckets (vae_amd/parallel.py).  State-dict key names are the
reference's (networks.py:162-231).

Batch covariates (``VAE(batch_properties=...)``, modules.py:102-107): the first-layer weights of the encoders are
(H, G + n_cov).  Their rows are stored with a stride rounded up to 8 floats (``ld_enc_w``) so that the gene-wide GEMMs
keep 16-byte aligned rows for any n_cov; the parameter is a strided VIEW of shape (H, G + n_cov), the padding columns
are zero in the weights and in the gradients and stay so (nothing reads or writes them; AdamW maps 0 -> 0).
"""
from __future__ import annotations

from collections import OrderedDict
from typing import Dict, Iterable, List, Optional, Tuple

import torch


def _enc(prefix: str, n: int, what: str) -> str:
    return f"{prefix}.encoder.fc_layers.{4 * n + (0 if what in ('w', 'b') else 2)}." + (
        "weight" if what in ("w", "g") else "bias")


def _dec(n: int, what: str) -> str:
    return f"decoder.px_decoder.fc_layers.{4 * n + (0 if what in ('w', 'b') else 2)}." + (
        "weight" if what in ("w", "g") else "bias")


class ParamArena:
    """Owns the flat parameter and gradient buffers and hands out views."""

    def __init__(self, params: Dict[str, torch.Tensor], n_layers: int, likelihood: str, device):
        self.n_layers = n_layers
        self.likelihood = likelihood
        self.device = torch.device(device)
        self.version = 0          # bumped by touch(): the parameters were written behind the fused optimiser's back
        names = set(params.keys())
        cell = sorted(k for k in names if k.startswith("cell_decoder."))
        # groups = runs of parameters that the kernels address as ONE stacked matrix / vector;
        # only group starts are aligned to 16 bytes so that members stay adjacent
        big = [[_enc("z_encoder", 0, "w"), _enc("l_encoder", 0, "w")],
               ["decoder.px_scale_decoder.0.weight", "decoder.px_dropout_decoder.weight"]]
        mid: List[List[str]] = [[_enc("z_encoder", n, "w")] for n in range(1, n_layers)]
        mid += [["z_encoder.mean_encoder.weight", "z_encoder.var_encoder.weight"],
                ["l_encoder.mean_encoder.weight", "l_encoder.var_encoder.weight"]]
        mid += [[_dec(n, "w")] for n in range(n_layers)]
        small: List[List[str]] = [["px_r"], [_enc("z_encoder", 0, "b"), _enc("l_encoder", 0, "b")],
                                  ["decoder.px_scale_decoder.0.bias", "decoder.px_dropout_decoder.bias"],
                                  ["z_encoder.mean_encoder.bias", "z_encoder.var_encoder.bias"],
                                  ["l_encoder.mean_encoder.bias", "l_encoder.var_encoder.bias"]]
        for n in range(n_layers):
            small += [[_enc("z_encoder", n, "g")], [_enc("z_encoder", n, "beta")]]
            if n > 0:
                small.append([_enc("z_encoder", n, "b")])
            small += [[_dec(n, "b")], [_dec(n, "g")], [_dec(n, "beta")]]
        small += [[_enc("l_encoder", 0, "g")], [_enc("l_encoder", 0, "beta")]]
        small += [[k] for k in cell]
        unused = [["decoder.px_r_decoder.weight"], ["decoder.px_r_decoder.bias"]]
        groups = big + mid + small + unused
        order = [k for grp in groups for k in grp]
        missing = [k for k in order if k not in names]
        extra = names - set(order)
        if missing or extra:
            raise KeyError(f"ParamArena: parameter set mismatch; missing {missing}, unexpected {sorted(extra)}")

        self.offsets: "OrderedDict[str, Tuple[int, torch.Size]]" = OrderedDict()
        G = params["px_r"].shape[0]
        self.n_cov = int(params[big[0][0]].shape[1]) - G          # one-hot batch-covariate columns (0: none)
        if self.n_cov < 0 or params[big[0][1]].shape[1] != G + self.n_cov or params[_dec(0, "w")].shape[1] != \
                params["z_encoder.mean_encoder.weight"].shape[0] + self.n_cov:
            raise ValueError("ParamArena: first-layer weight widths do not agree on the number of covariate columns")
        self.ld_enc_w = (G + self.n_cov + 7) & ~7 if self.n_cov else G
        self.row_stride: Dict[str, int] = {k: self.ld_enc_w for k in big[0]} if self.n_cov else {}
        off = 0
        for grp in groups:
            off = (off + 3) & ~3  # 16-byte alignment of every group start (vector loads)
            if grp is small[0]:
                self.small_begin = off
            if grp is unused[0]:
                self.small_end = off
            for k in grp:
                self.offsets[k] = (off, params[k].shape)
                off += self.extent(k)
        self.used_end = self.small_end
        self.cell_begin = self.offsets[cell[0]][0] if cell else self.small_end   # cell heads: last groups of "small"
        self.total = (off + 3) & ~3
        self.flat = torch.zeros(self.total, dtype=torch.float32, device=self.device)
        self.flat_grad = torch.zeros(self.total, dtype=torch.float32, device=self.device)
        for k in order:
            self.p(k).copy_(params[k].detach().to(device=self.device, dtype=torch.float32))
        # parameters the loss never reaches (grad None in the reference): SURVEY quirk 6
        self.no_grad = {k for grp in unused for k in grp}
        if likelihood == "nb":
            self.no_grad |= {"decoder.px_dropout_decoder.weight", "decoder.px_dropout_decoder.bias"}

    # -- views ---------------------------------------------------------------------------
    def touch(self) -> None:
        """Tell dependants that the parameters were written from OUTSIDE the fused optimiser (a checkpoint load, a
        broadcast, a manual re-initialisation): side products kept by ``FusedAdamW`` (weight maxima, bf16 weight images)
        are refreshed before the next step uses them."""
        self.version += 1

    def extent(self, k: str) -> int:
        """Floats the parameter occupies in the arena (rows x row stride for a padded matrix)."""
        _, shape = self.offsets[k]
        if k in self.row_stride:
            return shape[0] * self.row_stride[k]
        n = 1
        for s in shape:
            n *= s
        return n

    def _view(self, buf: torch.Tensor, k: str) -> torch.Tensor:
        off, shape = self.offsets[k]
        if k in self.row_stride:
            return buf.as_strided(tuple(shape), (self.row_stride[k], 1), off)
        return buf[off:off + self.extent(k)].view(shape)

    def p(self, k: str) -> torch.Tensor:
        return self._view(self.flat, k)

    def g(self, k: str) -> torch.Tensor:
        return self._view(self.flat_grad, k)

    def _cat(self, buf: torch.Tensor, keys: Iterable[str]) -> torch.Tensor:
        """One tensor spanning adjacent parameters stacked along dim 0."""
        keys = list(keys)
        off0, shape0 = self.offsets[keys[0]]
        cols = shape0[1:] if len(shape0) > 1 else torch.Size([])
        rows, expect = 0, off0
        for k in keys:
            off, shape = self.offsets[k]
            tail = shape[1:] if len(shape) > 1 else torch.Size([])
            if off != expect or tail != cols or self.row_stride.get(k) != self.row_stride.get(keys[0]):
                raise RuntimeError(f"ParamArena: {keys} are not adjacent / compatible in the arena")
            expect = off + self.extent(k)
            rows += shape[0]
        if keys[0] in self.row_stride:
            return buf.as_strided((rows,) + tuple(cols), (self.row_stride[keys[0]], 1), off0)
        return buf[off0:expect].view((rows,) + tuple(cols))

    def pcat(self, *keys: str) -> torch.Tensor:
        return self._cat(self.flat, keys)

    def gcat(self, *keys: str) -> torch.Tensor:
        return self._cat(self.flat_grad, keys)

    def zero_cell_grads(self) -> None:
        """The cell-head gradients come from torch autograd, which ACCUMULATES: clear them once per step."""
        if self.cell_begin < self.small_end:
            self.flat_grad[self.cell_begin:self.small_end].zero_()

    def zero_small_grads(self) -> None:
        self.flat_grad[self.small_begin:self.small_end].zero_()

    def grads(self) -> Dict[str, Optional[torch.Tensor]]:
        return {k: (None if k in self.no_grad else self.g(k)) for k in self.offsets}

    def state_dict(self) -> Dict[str, torch.Tensor]:
        return {k: self.p(k) for k in self.offsets}
