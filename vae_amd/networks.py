"""VAE container with the reference's constructor and parameter names (src/networks.py:96-231).

Sub-modules are created in the reference's order (px_r, z_encoder, cell_decoder, l_encoder,
decoder) with the same torch initialisers, so ``torch.manual_seed(s); VAE(...)`` yields the
reference's initial weights and the state_dict keys of SURVEY.md §3.4.
"""
from __future__ import annotations

from typing import Any, Dict, Optional

import torch
import torch.nn as nn

from . import ops
from .modules import CellDecoder, DecoderSCVI, Encoder, GroupedEncoder

_DISPERSIONS = ("gene", "gene-batch", "gene-label", "gene-cell")


class VAE(nn.Module):
    def __init__(
        self,
        n_input: int,
        n_batch: int = 0,
        n_labels: int = 0,
        n_hidden: int = 128,
        n_hidden_decoder: int = 128,
        n_hidden_library: int = 128,
        n_latent: int = 10,
        n_latent_cell_decoder: int = 10,
        n_layers: int = 1,
        dropout_rate: float = 0.1,
        input_dropout_rate: float = 0.0,
        cell_properties: Optional[Dict[str, Any]] = None,
        batch_properties: Optional[Dict[str, Any]] = None,
        dispersion: str = "gene",
        log_variational: bool = True,
        reconstruction_loss: str = "zinb",
        latent_distribution: str = "normal",
        grouped_encoder: bool = False,
        grad_reverse_dict: Optional[Dict] = None,
        cell_decoder_hidden_layer: bool = False,
    ):
        super().__init__()
        if dispersion not in _DISPERSIONS:
            raise ValueError(f"dispersion must be one of {list(_DISPERSIONS)}, but input was {dispersion}.")
        self.dispersion = dispersion
        self.n_input = n_input
        self.n_latent = n_latent
        self.n_layers = n_layers
        self.log_variational = log_variational
        self.reconstruction_loss = reconstruction_loss
        self.n_batch = n_batch
        self.n_labels = n_labels
        self.latent_distribution = latent_distribution
        self.grouped_encoder = grouped_encoder

        px_r_cols = {"gene": (), "gene-batch": (n_batch,), "gene-label": (n_labels,)}
        if dispersion in px_r_cols:
            self.px_r = torch.nn.Parameter(torch.randn(n_input, *px_r_cols[dispersion]))

        n_properties = len(cell_properties)  # the reference requires a dict here too (networks.py:179)
        enc_kw = dict(batch_properties=batch_properties, n_layers=n_layers, n_hidden=n_hidden,
                      dropout_rate=dropout_rate, input_dropout_rate=input_dropout_rate)
        if grouped_encoder:
            self.z_encoder = GroupedEncoder(n_input, n_latent, cell_properties, latent_distribution=latent_distribution,
                                            **enc_kw)
        else:
            self.z_encoder = Encoder(n_input, n_latent, distribution=latent_distribution, **enc_kw)
        if cell_properties is not None:
            self.cell_decoder = CellDecoder(n_latent_cell_decoder, cell_properties, grad_reverse_dict=grad_reverse_dict,
                                            use_hidden_layer=cell_decoder_hidden_layer)
        # library encoder: always one layer, no dropout (networks.py:215-222)
        self.l_encoder = Encoder(n_input, 1, batch_properties=batch_properties, n_layers=1, n_hidden=n_hidden_library,
                                 dropout_rate=0.0)
        self.decoder = DecoderSCVI(n_latent * n_properties if grouped_encoder else n_latent, n_input, n_layers=n_layers,
                                   n_hidden=n_hidden_decoder, batch_properties=batch_properties,
                                   dropout_rate=dropout_rate)

    def set_gemm_precision(self, precision: int) -> None:
        """ops.FP32 (fp32-grade: operands split exactly into three bf16 terms, six bf16 MFMA products, fp32 accumulate)
        or ops.BF16 (operands rounded to bf16, fp32 accumulate) for every Linear of the module path."""
        for m in self.modules():
            if hasattr(m, "precision"):
                m.precision = precision

    def forward(self, x, batch_labels=None, batch_mask=None):
        """Encoders + decoder in the live order of LitVAE.training_step (task.py:460-467).  (The
        reference's own VAE.forward is stale and raises)"""
        qz_m, qz_v, z = self.z_encoder(x, batch_labels, batch_mask)
        ql_m, ql_v, library = self.l_encoder(x, batch_labels, batch_mask)
        px_scale, px_r, px_rate, px_dropout = self.decoder(self.dispersion, z, library, batch_labels, batch_mask)
        return dict(qz_m=qz_m, qz_v=qz_v, z=z, ql_m=ql_m, ql_v=ql_v, library=library, px_scale=px_scale, px_r=px_r,
                    px_rate=px_rate, px_dropout=px_dropout)


def load_model(model_save_path, model):
    """Size-matched state_dict load that strips a leading 'network.' and skips cell-property
    heads (networks.py:542-568).  Only weights_only checkpoints are read."""
    ckpt = torch.load(model_save_path, map_location="cpu", weights_only=True)
    state = ckpt.get("state_dict", ckpt.get("model_state_dict", ckpt))
    own = model.state_dict()
    for k, v in state.items():
        k = k[len("network."):] if k.startswith("network.") else k
        if "cell_property" in k or k not in own or own[k].shape != v.shape:
            continue
        own[k].copy_(v)
    return model
