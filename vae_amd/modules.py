"""The reference's nn.Module API (src/modules.py) on the HIP kernels.

Constructor signatures, sub-module names (hence state_dict keys) and forward signatures are
the reference's; the forwards run the GEMMs, GELU+LayerNorm, softmax/library scaling on the
hand-written gfx950 kernels through vae_amd.autograd.  Tiny (B, latent) element-wise steps
(softplus of the one variance column, the rsample affine) stay torch ops on the GPU.
"""
from __future__ import annotations

from typing import Any, Dict, Optional

import numpy as np
import torch
import torch.nn.functional as F
from torch import nn

from . import ops
from .autograd import CountLinearFn, GeluLayerNormFn, LinearFn, SoftmaxRateFn
from .functional import GradReverseLayer


def reparameterize_gaussian(mu, var):
    """Normal(mu, sqrt(var)).rsample() (modules.py:11-12): mu + eps*sqrt(var), eps drawn with
    ``empty(shape).normal_()`` exactly as torch's rsample does (same generator consumption)."""
    scale = var.sqrt()
    shape = torch.broadcast_shapes(mu.shape, scale.shape)
    eps = torch.empty(shape, dtype=mu.dtype, device=mu.device).normal_()
    return mu + eps * scale


def identity(x):
    return x


def one_hot(index, n_cat):
    onehot = torch.zeros(index.size(0), n_cat, device=index.device)
    onehot.scatter_(1, index.type(torch.long), 1)
    return onehot.type(torch.float32)


def hip_linear(lin: nn.Linear, x: torch.Tensor, precision: int = ops.FP32) -> torch.Tensor:
    return LinearFn.apply(x, lin.weight, lin.bias, precision)


class LinearWithMask(nn.Module):
    """Masked linear layer of the pathways model (modules.py:25-42)."""

    def __init__(self, n_input: int, n_output: int, mask: torch.Tensor):
        super().__init__()
        self.W = nn.parameter.Parameter(data=torch.randn(n_input, n_output), requires_grad=True)
        self.b = nn.parameter.Parameter(data=torch.zeros((1, n_output), dtype=torch.float32), requires_grad=True)
        self.mask = mask

    def forward(self, x):
        W = (self.W * self.mask.to(x.device)).t().contiguous()
        return LinearFn.apply(x, W, self.b.view(-1), ops.FP32)


class FCLayers(nn.Module):
    """Per layer: Linear -> GELU(erf) -> LayerNorm(affine, eps 1e-5) -> Dropout (modules.py:74-132);
    optional masked one-hot batch covariates concatenated to the input (:123-130)."""

    def __init__(self, n_in: int, n_layers: int = 1, n_hidden: int = 128, dropout_rate: float = 0.1,
                 batch_properties: Optional[Dict[str, Any]] = None, layer_norm: bool = True):
        super().__init__()
        self.batch_dims = 0 if batch_properties is None else [
            len(batch_properties[k]["values"]) for k in batch_properties.keys()]
        self.act = nn.GELU()
        self.layer_norm = layer_norm
        self.n_layers = n_layers
        self.precision = ops.FP32
        layers_dim = [n_in + int(np.sum(self.batch_dims))] + n_layers * [n_hidden]
        layers = []
        for n in range(n_layers):
            layers.append(nn.Linear(layers_dim[n], layers_dim[n + 1]))
            layers.append(nn.GELU())
            if layer_norm:
                layers.append(nn.LayerNorm(n_hidden, elementwise_affine=True))
            layers.append(nn.Dropout(p=float(dropout_rate)))
        self.fc_layers = nn.Sequential(*layers)

    def _run(self, x: torch.Tensor, count_input: bool, cov: Optional[torch.Tensor] = None) -> torch.Tensor:
        mods = list(self.fc_layers)
        per = 4 if self.layer_norm else 3
        for n in range(self.n_layers):
            lin = mods[per * n]
            if n == 0 and count_input:
                # Encoder.forward normalises the counts BEFORE FCLayers appends the covariates (modules.py:264,267), so
                # with W = [W_x | W_c] the first layer is (x/mean - 1) W_x^T + cov W_c^T + b: the gene-wide product stays
                # on the count GEMM (normalisation fused into its operand staging), the few covariate columns add a
                # (B, n_cov) x (n_cov, H) product -- nothing 20k wide is concatenated or copied
                if cov is not None:
                    G = x.shape[1]
                    pre = CountLinearFn.apply(x, lin.weight[:, :G], lin.bias, self.precision)
                    pre = pre + LinearFn.apply(cov, lin.weight[:, G:], None, self.precision)
                else:
                    pre = CountLinearFn.apply(x, lin.weight, lin.bias, self.precision)
            else:
                pre = LinearFn.apply(x, lin.weight, lin.bias, self.precision)
            if self.layer_norm:
                ln = mods[per * n + 2]
                x = GeluLayerNormFn.apply(pre, ln.weight, ln.bias)
            else:
                x = F.gelu(pre)
            x = mods[per * n + per - 1](x)  # Dropout (identity in eval or at p = 0)
        return x

    def forward(self, x: torch.Tensor, batch_vals: torch.Tensor, batch_mask: torch.Tensor, _count_input: bool = False):
        cov = None
        if np.sum(self.batch_dims) > 0:
            batch_vals[batch_vals < 0] = 0                      # in place, as the reference (modules.py:124)
            batch_vars = []
            for n in range(len(self.batch_dims)):
                b = F.one_hot(batch_vals[:, n], num_classes=self.batch_dims[n])
                batch_vars.append(b * batch_mask[:, n: n + 1])
            if _count_input:
                cov = torch.cat(batch_vars, dim=-1).to(torch.float32)
            else:
                x = torch.cat((x, *batch_vars), dim=-1)
        return self._run(x, _count_input, cov)


class GeneDropout(nn.Module):
    """Input dropout whose per-gene keep probability follows how often the gene is expressed
    (role of modules.py:190-209).  An exponential moving average (``alpha``) of the fraction of
    cells with a non-zero count sets keep_g = clip(p * ema_g / mean(ema), 0, 0.9); kept entries are
    rescaled by 1 / mean(keep).  Identity in eval mode and for p < 1e-6 (every benchmark config)."""

    def __init__(self, p: float = 0.5, alpha: float = 0.95):
        super().__init__()
        self.p, self.alpha = p, alpha
        self.running_non_zero_means = 0.0   # plain attribute, not a buffer: absent from the state dict

    def _keep_probability(self, counts: torch.Tensor) -> torch.Tensor:
        expressed = counts.gt(0).float().mean(dim=0, keepdim=True)
        self.running_non_zero_means = self.alpha * self.running_non_zero_means + (1.0 - self.alpha) * expressed
        ema = self.running_non_zero_means
        return (self.p * ema / ema.mean()).clamp(0.0, 0.9)

    def forward(self, input: torch.Tensor) -> torch.Tensor:
        if self.p < 1e-6 or not self.training:
            return input
        keep = self._keep_probability(input)
        u = torch.rand(*input.size()).to(device=input.device)   # drawn on the host like the reference (same RNG stream)
        return input * (u < keep).float() / keep.mean()


class Encoder(nn.Module):
    """genes -> latent encoder (modules.py:213-271): x/mean-1 -> FCLayers -> mean / ONE shared
    variance head -> softplus + 1e-4 -> reparameterised sample."""

    def __init__(self, n_input: int, n_output: int, batch_properties: Optional[Dict[str, Any]] = None, n_layers: int = 1,
                 n_hidden: int = 128, dropout_rate: float = 0.1, input_dropout_rate: float = 0.0,
                 distribution: str = "normal", shared_variance: bool = True):
        super().__init__()
        self.distribution = distribution
        self.encoder = FCLayers(n_in=n_input, n_layers=n_layers, n_hidden=n_hidden, dropout_rate=dropout_rate,
                                batch_properties=batch_properties)
        self.mean_encoder = nn.Linear(n_hidden, n_output)
        self.var_encoder = nn.Linear(n_hidden, 1) if shared_variance else nn.Linear(n_hidden, n_output)
        self.ln = nn.LayerNorm(n_input, elementwise_affine=False)  # parameter-free; unused on the live path
        self.z_transformation = nn.Softmax(dim=-1) if distribution == "ln" else identity
        self.drop = GeneDropout(p=input_dropout_rate)
        self.softplus = nn.Softplus()

    def forward(self, x: torch.Tensor, batch_labels: torch.Tensor, batch_mask: torch.Tensor):
        x = self.drop(x)
        q = self.encoder(x, batch_labels, batch_mask, _count_input=True)  # normalisation fused into the GEMM
        q_m = hip_linear(self.mean_encoder, q, self.encoder.precision)
        q_v = self.softplus(hip_linear(self.var_encoder, q, self.encoder.precision)) + 1e-4
        latent = self.z_transformation(reparameterize_gaussian(q_m, q_v))
        return q_m, q_v, latent


class GroupedEncoder(nn.Module):
    """A separate Encoder per cell property; means, variances and samples are concatenated along the
    latent axis in the order of ``cell_properties`` (role of modules.py:134-188)."""

    def __init__(self, n_input: int, n_latent: int, cell_properties: Optional[Dict[str, Any]] = None,
                 batch_properties: Optional[Dict[str, Any]] = None, n_layers: int = 1, n_hidden: int = 128,
                 dropout_rate: float = 0.1, input_dropout_rate: float = 0.0, latent_distribution: str = "normal"):
        super().__init__()
        self.z_enc = nn.ModuleDict()
        self.cell_properties = cell_properties
        self.cell_predict = nn.ModuleDict()   # empty in the reference too; kept for module-tree parity
        enc_args = dict(batch_properties=batch_properties, n_layers=n_layers, n_hidden=n_hidden, dropout_rate=dropout_rate,
                        input_dropout_rate=input_dropout_rate, distribution=latent_distribution)
        for name in cell_properties:
            self.z_enc[name] = Encoder(n_input, n_latent, **enc_args)

    def forward(self, x: torch.Tensor, batch_labels: torch.Tensor, batch_mask: torch.Tensor):
        parts = [self.z_enc[name](x, batch_labels, batch_mask) for name in self.cell_properties]
        q_m, q_v, latent = (torch.cat(group, dim=-1) for group in zip(*parts))
        return q_m, q_v, latent


class CellDecoder(nn.Module):
    """Cell-property heads on the latent (role of modules.py:274-324).  A property listed in
    ``grad_reverse_dict`` sees the latent through a gradient-reversal layer (adversarial head), a
    ``stop_grad`` property sees it detached, the others see it as is.  Tiny: B x L x n_targets."""

    def __init__(self, latent_dim: int, cell_properties: Dict[str, Any], grad_reverse_dict: Optional[Dict] = None,
                 use_hidden_layer: bool = True):
        super().__init__()
        self.latent_dim = latent_dim
        self.cell_properties = cell_properties
        self.cell_mlp = nn.ModuleDict()
        self.grad_reverse_dict = grad_reverse_dict
        if grad_reverse_dict is not None:
            self.grad_reverse = nn.ModuleDict({k: GradReverseLayer(lam) for k, lam in grad_reverse_dict.items()})
        for name, prop in cell_properties.items():   # creation order = RNG order = state-dict order
            width = len(prop["values"]) if prop["discrete"] else 1
            if not use_hidden_layer:
                head = nn.Linear(latent_dim, width, bias=True)
            else:
                head = nn.Sequential(nn.Linear(latent_dim, 2 * latent_dim), nn.GELU(), nn.Linear(2 * latent_dim, width))
            self.cell_mlp[name] = head

    def _view_of_latent(self, name: str, prop: Dict[str, Any], latent: torch.Tensor) -> torch.Tensor:
        if name in self.grad_reverse_dict.keys():   # a None dict raises here, as the reference does (:317)
            return self.grad_reverse[name](latent)
        return latent.detach() if prop["stop_grad"] else latent

    def forward(self, latent: torch.Tensor):
        return {name: self.cell_mlp[name](self._view_of_latent(name, prop, latent))
                for name, prop in self.cell_properties.items()}


class DecoderSCVI(nn.Module):
    """latent -> genes decoder (modules.py:328-406): FCLayers, then the px_scale (softmax over
    genes), px_dropout and (gene-cell only) px_r heads; px_rate = exp(library) * px_scale."""

    def __init__(self, n_input: int, n_output: int, n_layers: int = 1, n_hidden: int = 128, dropout_rate: float = 0.1,
                 batch_properties: Optional[Dict[str, Any]] = None):
        super().__init__()
        self.px_decoder = FCLayers(n_in=n_input, n_layers=n_layers, n_hidden=n_hidden, dropout_rate=dropout_rate,
                                   batch_properties=batch_properties)
        self.px_scale_decoder = nn.Sequential(nn.Linear(n_hidden, n_output), nn.Softmax(dim=-1))
        self.px_r_decoder = nn.Linear(n_hidden, n_output)
        self.px_dropout_decoder = nn.Linear(n_hidden, n_output)

    def forward(self, dispersion: str, z: torch.Tensor, library: torch.Tensor, batch_labels: torch.Tensor,
                batch_mask: torch.Tensor):
        P = self.px_decoder.precision
        px = self.px_decoder(z, batch_labels, batch_mask)
        logits = hip_linear(self.px_scale_decoder[0], px, P)
        px_scale, px_rate = SoftmaxRateFn.apply(logits, library)
        px_dropout = hip_linear(self.px_dropout_decoder, px, P)
        px_r = hip_linear(self.px_r_decoder, px, P) if dispersion == "gene-cell" else None
        return px_scale, px_r, px_rate, px_dropout
