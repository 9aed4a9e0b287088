"""CPU oracle for the scVI-style VAE training hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``vae_amd/`` may import this file; only
``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg do,
and only as the checker / the reported CPU baseline -- never as the thing shipped.

It is a plain-PyTorch (CPU, fp32 or fp64) restatement -- written from the
reference's behaviour, not copied -- of the live training-step sequence of the
reference (``/root/reference/src/task.py:453-502``), expressed functionally over a
flat ``{state_dict key: tensor}`` parameter dict so that it has no dependency on
the product package.  Each function cites the reference lines it follows.

Pinning: ``tests/golden/*.npz`` were produced by importing the *reference*
modules (``tests/golden/make_goldens.py``, run once in the build container where
``/root/reference`` is mounted); ``tests/test_oracle_golden.py`` asserts that this
restatement reproduces them.  The reference itself never travels to the GPU box.

bf16 mode (``gemm_dtype=torch.bfloat16``): defined by this project (SURVEY.md §5,
"Mixed precision") as: both operands of every Linear GEMM are rounded to bf16
(round-to-nearest-even), products accumulate in fp32, every other op is fp32.
``logits_bf16=True`` adds what the reference's autocast(bfloat16) does to the two gene-sized decoder heads
(config.py:87: an nn.Linear under autocast RETURNS a bf16 tensor): their outputs (accumulator + bias) are rounded to
bf16 before the softmax / the likelihood read them; the gradient passes the rounding unchanged (as a dtype cast does
in autograd).  The product path stores those logits as bf16 for shapes on its tile pipeline.
"""
from __future__ import annotations

import math
from typing import Dict, Optional

import torch
import torch.nn.functional as F

Tensor = torch.Tensor
EPS = 1e-8


# --------------------------------------------------------------------------- #
# likelihoods  (reference: src/distributions.py:22-62 and :65-90)
# --------------------------------------------------------------------------- #
def log_nb_positive(x: Tensor, mu: Tensor, theta: Tensor, eps: float = EPS) -> Tensor:
    """Per-element NB log-likelihood; distributions.py:65-90."""
    if theta.dim() == 1:
        theta = theta.unsqueeze(0)
    l_tm = torch.log(theta + mu + eps)
    return (
        theta * (torch.log(theta + eps) - l_tm)
        + x * (torch.log(mu + eps) - l_tm)
        + torch.lgamma(x + theta)
        - torch.lgamma(theta)
        - torch.lgamma(x + 1)
    )


def log_zinb_positive(x: Tensor, mu: Tensor, theta: Tensor, pi: Tensor, eps: float = EPS) -> Tensor:
    """Per-element ZINB log-likelihood; distributions.py:22-62."""
    if theta.dim() == 1:
        theta = theta.unsqueeze(0)
    sp_pi = F.softplus(-pi)
    l_t = torch.log(theta + eps)
    l_tm = torch.log(theta + mu + eps)
    ptl = -pi + theta * (l_t - l_tm)
    zero_case = F.softplus(ptl) - sp_pi
    nz_case = (
        -sp_pi + ptl + x * (torch.log(mu + eps) - l_tm)
        + torch.lgamma(x + theta) - torch.lgamma(theta) - torch.lgamma(x + 1)
    )
    is_zero = (x < eps).to(zero_case.dtype)
    is_pos = (x > eps).to(zero_case.dtype)
    return is_zero * zero_case + is_pos * nz_case


# --------------------------------------------------------------------------- #
# building blocks
# --------------------------------------------------------------------------- #
def _linear(a: Tensor, w: Tensor, b: Optional[Tensor], gemm_dtype) -> Tensor:
    """nn.Linear; in bf16 mode both GEMM operands are rounded to bf16, fp32 accumulate."""
    if gemm_dtype is torch.bfloat16:
        a = a.to(torch.bfloat16).to(w.dtype if w.dtype != torch.bfloat16 else torch.float32)
        w = w.to(torch.bfloat16).to(a.dtype)
    y = a @ w.t()
    return y if b is None else y + b


def batch_covariates(batch_labels: Tensor, batch_mask: Tensor, sizes) -> Tensor:
    """The masked one-hot block FCLayers.forward appends to its input (modules.py:123-130): labels below 0 count as 0
    (the reference clamps them in place), every property's one-hot is multiplied by its mask column."""
    lab = batch_labels.clamp(min=0)
    cols = [F.one_hot(lab[:, n], num_classes=int(k)) * batch_mask[:, n:n + 1] for n, k in enumerate(sizes)]
    return torch.cat(cols, dim=-1)


def fc_layers(p: Dict[str, Tensor], prefix: str, x: Tensor, n_layers: int, gemm_dtype=None, masks=None, cov=None) -> Tensor:
    """FCLayers; modules.py:110-115,121-132.  ``cov``: the batch-covariate block, concatenated to the input.

    Per layer n: Linear (index 4n) -> GELU(erf) -> LayerNorm(eps 1e-5, affine; index 4n+2) -> Dropout
    (index 4n+3).  ``masks[n]`` is the dropout MULTIPLIER of layer n (keep / (1 - p), what nn.Dropout
    multiplies by in training mode); None = dropout off / eval.
    """
    if cov is not None:
        x = torch.cat((x, cov.to(x.dtype)), dim=-1)
    for n in range(n_layers):
        w = p[f"{prefix}.fc_layers.{4 * n}.weight"]
        b = p[f"{prefix}.fc_layers.{4 * n}.bias"]
        g = p[f"{prefix}.fc_layers.{4 * n + 2}.weight"]
        be = p[f"{prefix}.fc_layers.{4 * n + 2}.bias"]
        x = _linear(x, w, b, gemm_dtype)
        x = F.gelu(x)
        x = F.layer_norm(x, (w.shape[0],), g, be, 1e-5)
        if masks is not None and masks[n] is not None:
            x = x * masks[n].to(x.dtype)
    return x


def encoder_forward(p, prefix: str, x: Tensor, noise: Tensor, n_layers: int, gemm_dtype=None, masks=None, cov=None):
    """Encoder.forward, modules.py:261-271 (GeneDropout identity, z_transformation identity).

    ``noise`` is the standard-normal draw Normal.rsample would make (modules.py:11-12);
    q_v has ONE column (shared_variance, modules.py:249) and broadcasts over the latent.
    """
    xn = x / x.mean(dim=-1, keepdim=True) - 1
    q = fc_layers(p, f"{prefix}.encoder", xn, n_layers, gemm_dtype, masks, cov)   # covariates join AFTER the normalisation
    q_m = _linear(q, p[f"{prefix}.mean_encoder.weight"], p[f"{prefix}.mean_encoder.bias"], gemm_dtype)
    q_v = F.softplus(_linear(q, p[f"{prefix}.var_encoder.weight"], p[f"{prefix}.var_encoder.bias"], gemm_dtype)) + 1e-4
    latent = q_m + noise * torch.sqrt(q_v)
    return q_m, q_v, latent


def _as_bf16_tensor(y: Tensor) -> Tensor:
    """Value of y after a round trip through a bf16 tensor; gradient = identity (what a dtype cast back-propagates)."""
    return y + (y.detach().to(torch.bfloat16).to(y.dtype) - y.detach())


def decoder_forward(p, z: Tensor, library: Tensor, n_layers: int, gemm_dtype=None, masks=None, cov=None, logits_bf16=False):
    """DecoderSCVI.forward for dispersion != 'gene-cell'; modules.py:371-406."""
    px = fc_layers(p, "decoder.px_decoder", z, n_layers, gemm_dtype, masks, cov)
    logit_scale = _linear(px, p["decoder.px_scale_decoder.0.weight"], p["decoder.px_scale_decoder.0.bias"], gemm_dtype)
    px_dropout = _linear(px, p["decoder.px_dropout_decoder.weight"], p["decoder.px_dropout_decoder.bias"], gemm_dtype)
    if logits_bf16:
        logit_scale, px_dropout = _as_bf16_tensor(logit_scale), _as_bf16_tensor(px_dropout)
    px_scale = torch.softmax(logit_scale, dim=-1)
    px_rate = torch.exp(library) * px_scale
    return px_scale, px_rate, px_dropout


def decoder_px_r(p, z: Tensor, n_layers: int, gemm_dtype=None, masks=None) -> Tensor:
    """The third decoder head of dispersion 'gene-cell' (modules.py:366,405): px_r = Linear(px), one
    dispersion per cell and gene (theta = exp(px_r), task.py:483)."""
    px = fc_layers(p, "decoder.px_decoder", z, n_layers, gemm_dtype, masks)
    return _linear(px, p["decoder.px_r_decoder.weight"], p["decoder.px_r_decoder.bias"], gemm_dtype)


def gaussian_kl(qm: Tensor, qv: Tensor, pm, pv) -> Tensor:
    """KL(N(qm, sqrt(qv)) || N(pm, sqrt(pv))) per element, the closed form torch's
    ``_kl_normal_normal`` evaluates (task.py:190-198)."""
    sq, sp = torch.sqrt(qv), torch.sqrt(torch.as_tensor(pv, dtype=qm.dtype))
    var_ratio = (sq / sp) ** 2
    t1 = ((qm - pm) / sp) ** 2
    return 0.5 * (var_ratio + t1 - 1 - torch.log(var_ratio))


def cell_loss(p, z: Tensor, cell_properties: dict, grad_reverse: dict, targets: Tensor, mask: Tensor,
              gemm_dtype=None) -> Tensor:
    """CellDecoder.forward (use_hidden_layer False) + LitVAE._cell_loss;
    modules.py:312-324, task.py:419-451, functional.py:8-20.

    Gradient reversal is expressed as ``y = -lam*x + (1+lam)*x.detach()`` (value x, grad -lam).
    """
    total = z.new_zeros(())
    for n, (k, prop) in enumerate(cell_properties.items()):
        if k in grad_reverse:
            lam = float(grad_reverse[k])
            inp = -lam * z + (1.0 + lam) * z.detach()
        else:
            inp = z.detach() if prop["stop_grad"] else z
        pred = _linear(inp, p[f"cell_decoder.cell_mlp.{k}.weight"], p[f"cell_decoder.cell_mlp.{k}.bias"], gemm_dtype)
        if prop["discrete"]:
            per = F.cross_entropy(pred, targets[:, n].to(torch.int64), reduction="none", ignore_index=-100)
        else:
            per = (torch.squeeze(pred) - targets[:, n]) ** 2
        total = total + (per * mask[:, n]).mean()
    return total


# --------------------------------------------------------------------------- #
# the training step (task.py:453-502 with gene_loss task.py:171-211)
# --------------------------------------------------------------------------- #
def training_step(
    p: Dict[str, Tensor],
    x: Tensor,
    eps_z: Tensor,
    eps_l: Tensor,
    *,
    n_layers: int = 1,
    likelihood: str = "zinb",
    l_mean: float = 0.0,
    l_var: float = 1.0,
    kl_weight: float = 1.0,
    gene_loss_coeff: float = 5e-3,
    cell_properties: Optional[dict] = None,
    grad_reverse: Optional[dict] = None,
    cell_targets: Optional[Tensor] = None,
    cell_mask: Optional[Tensor] = None,
    gemm_dtype=None,
    drop_masks: Optional[dict] = None,
    x_z: Optional[Tensor] = None,
    dispersion: str = "gene",
    cov: Optional[Tensor] = None,
    logits_bf16: bool = False,
) -> Dict[str, Tensor]:
    """Forward of one training step; returns every intermediate the goldens pin.

    Training-mode regularisers enter as DATA, so that the step stays a deterministic function:
    ``drop_masks = {"z_encoder": [m_0, ...], "decoder": [m_0, ...]}`` are the nn.Dropout multipliers
    of each hidden layer (modules.py:115; the library encoder is built with dropout 0,
    networks.py:215-222); ``x_z`` is the z-encoder's input after GeneDropout (modules.py:190-209,263)
    -- the library encoder and the likelihood see ``x`` itself.  ``cov``: ``batch_covariates(...)`` of a model built
    with ``batch_properties`` (it enters both encoders and the decoder, modules.py:121-130)."""
    dm = drop_masks or {}
    qz_m, qz_v, z = encoder_forward(p, "z_encoder", x if x_z is None else x_z, eps_z, n_layers, gemm_dtype,
                                    dm.get("z_encoder"), cov)
    ql_m, ql_v, library = encoder_forward(p, "l_encoder", x, eps_l, 1, gemm_dtype, None, cov)  # networks.py:215-222
    px_scale, px_rate, px_dropout = decoder_forward(p, z, library, n_layers, gemm_dtype, dm.get("decoder"), cov, logits_bf16)
    if dispersion == "gene":
        theta = torch.exp(p["px_r"])  # task.py:481-483
    elif dispersion == "gene-cell":
        theta = torch.exp(decoder_px_r(p, z, n_layers, gemm_dtype, dm.get("decoder")))  # modules.py:405, task.py:483
    else:
        raise ValueError(dispersion)  # 'gene-batch' / 'gene-label' reference undefined names (task.py:475-480)

    if likelihood == "poisson":  # task.py:167-168: torch Poisson.log_prob = xlogy(x, rate) - rate - lgamma(x + 1)
        ll = torch.xlogy(x, px_rate) - px_rate - torch.lgamma(x + 1)
    elif likelihood == "zinb":
        ll = log_zinb_positive(x, px_rate, theta, px_dropout)
    elif likelihood == "nb":
        ll = log_nb_positive(x, px_rate, theta)
    else:
        raise ValueError(likelihood)
    recon = -ll.sum(dim=-1)  # task.py:156-165

    kl_z = gaussian_kl(qz_m, qz_v, 0.0, 1.0).sum(dim=1)
    kl_l = gaussian_kl(ql_m, ql_v, l_mean, l_var).sum(dim=1)
    gene_loss = recon.mean() + kl_weight * (kl_l + kl_z).mean()  # task.py:202

    if cell_properties:
        c_loss = cell_loss(p, z, cell_properties, grad_reverse or {}, cell_targets, cell_mask, gemm_dtype)
    else:
        c_loss = x.new_zeros(())
    loss = gene_loss_coeff * gene_loss + c_loss  # task.py:489
    return dict(
        qz_m=qz_m, qz_v=qz_v, z=z, ql_m=ql_m, ql_v=ql_v, library=library,
        px_scale=px_scale, px_rate=px_rate, px_dropout=px_dropout,
        recon=recon, kl_z=kl_z, kl_l=kl_l, gene_loss=gene_loss, cell_loss=c_loss, loss=loss,
    )


def training_step_with_grads(p: Dict[str, Tensor], *args, **kw):
    """Runs ``training_step`` on leaf copies of ``p`` and back-propagates ``loss``.

    Returns ``(outputs, grads)``; ``grads[name]`` is ``None`` for parameters the loss
    does not reach (px_r_decoder.*, px_dropout_decoder.* under NB, unused heads)."""
    leaves = {k: v.detach().clone().requires_grad_(True) for k, v in p.items()}
    out = training_step(leaves, *args, **kw)
    out["loss"].backward()
    grads = {k: (None if v.grad is None else v.grad.detach()) for k, v in leaves.items()}
    return {k: v.detach() for k, v in out.items()}, grads


# --------------------------------------------------------------------------- #
# synthetic model / data used by tests, smoke() and bench.py's cpu_baseline
# --------------------------------------------------------------------------- #
def init_params(G: int, H: int, Hd: int, Hl: int, L: int, n_layers: int, *, seed: int = 43,
                cell_properties: Optional[dict] = None, dtype=torch.float32) -> Dict[str, Tensor]:
    """Random parameters with torch's default Linear/LayerNorm init distributions and the
    state_dict key names of the reference VAE (networks.py:162-231)."""
    g = torch.Generator().manual_seed(seed)

    def lin(name, n_out, n_in, out):
        bound = 1.0 / math.sqrt(n_in)
        out[f"{name}.weight"] = (torch.rand(n_out, n_in, generator=g, dtype=torch.float64) * 2 - 1).mul(bound).to(dtype)
        out[f"{name}.bias"] = (torch.rand(n_out, generator=g, dtype=torch.float64) * 2 - 1).mul(bound).to(dtype)

    def ln(name, n, out):
        # perturbed away from (1, 0) so that affine-gradient bugs are visible
        out[f"{name}.weight"] = (1 + 0.1 * torch.randn(n, generator=g, dtype=torch.float64)).to(dtype)
        out[f"{name}.bias"] = (0.1 * torch.randn(n, generator=g, dtype=torch.float64)).to(dtype)

    p: Dict[str, Tensor] = {}
    p["px_r"] = torch.randn(G, generator=g, dtype=torch.float64).to(dtype)
    for n in range(n_layers):
        lin(f"z_encoder.encoder.fc_layers.{4 * n}", H, G if n == 0 else H, p)
        ln(f"z_encoder.encoder.fc_layers.{4 * n + 2}", H, p)
    lin("z_encoder.mean_encoder", L, H, p)
    lin("z_encoder.var_encoder", 1, H, p)
    for k, prop in (cell_properties or {}).items():
        n_t = len(prop["values"]) if prop["discrete"] else 1
        lin(f"cell_decoder.cell_mlp.{k}", n_t, L, p)
    lin("l_encoder.encoder.fc_layers.0", Hl, G, p)
    ln("l_encoder.encoder.fc_layers.2", Hl, p)
    lin("l_encoder.mean_encoder", 1, Hl, p)
    lin("l_encoder.var_encoder", 1, Hl, p)
    for n in range(n_layers):
        lin(f"decoder.px_decoder.fc_layers.{4 * n}", Hd, L if n == 0 else Hd, p)
        ln(f"decoder.px_decoder.fc_layers.{4 * n + 2}", Hd, p)
    lin("decoder.px_scale_decoder.0", G, Hd, p)
    lin("decoder.px_r_decoder", G, Hd, p)
    lin("decoder.px_dropout_decoder", G, Hd, p)
    return p


def synthetic_counts(B: int, G: int, *, seed: int = 43, zero_inflate: float = 0.0) -> Tensor:
    """Synthetic count matrix of SURVEY.md §8(d): x ~ Poisson(lambda_g * s_c), optional
    zero inflation, no all-zero rows (an all-zero row makes x/mean NaN)."""
    g = torch.Generator().manual_seed(seed)
    lam = torch.exp(torch.randn(G, generator=g) * 1.5 - 1.5).clamp(0, 50)
    s = torch.exp(torch.randn(B, 1, generator=g) * 0.25)
    x = torch.poisson(lam.unsqueeze(0) * s, generator=g)
    if zero_inflate > 0:
        x = x * (torch.rand(B, G, generator=g) >= zero_inflate).to(x.dtype)
    dead = x.sum(dim=1) == 0
    x[dead, 0] = 1.0
    return x.to(torch.float32)


def library_prior(x: Tensor):
    """Mean / variance of log1p(row sums) -- the constants data.py:233-244 hands to LitVAE."""
    s = torch.log1p(x.double().sum(dim=1))
    return float(s.mean()), float(s.var(unbiased=False))
