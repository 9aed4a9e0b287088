"""The fused VAE training step (forward + backward ELBO) on hand-written gfx950 kernels.

This is the hot path of the reference's ``LitVAE.training_step`` (task.py:453-502):
z/l encoders -> decoder -> exp(px_r) -> Gaussian KLs + NB/ZINB reconstruction ->
``loss = gene_loss_coeff * gene_loss + cell_loss`` -> gradients of every parameter.

The whole launch sequence lives in native code (csrc/step.hip): ONE C-ABI call enqueues the
forward, one the backward, on the current HIP stream, over a single pre-allocated workspace --
no autograd graph, no per-op Python, no allocation, no (B,G) temporaries beyond the logits
buffer (which its own gradient overwrites in place).  Parameters and gradients live in a
``ParamArena``.

Stage list (kernel in csrc/ -> reference lines):
  row_mean                x.mean(-1)                              modules.py:264
  gemm NT + x_mean        Linear(G, H+Hl) on x/mean-1             modules.py:111,264,267
  gelu_ln_fwd             GELU -> LayerNorm                       modules.py:112-114
  gemm NT                 [mean_encoder; var_encoder]             modules.py:268-269
  reparam_kl_fwd          softplus+1e-4, rsample, KL              modules.py:269-270, task.py:188-198
  gemm NT, gelu_ln_fwd    px_decoder                              modules.py:400
  gemm NT                 [px_scale_decoder.0; px_dropout_decoder] modules.py:401-402
  nbzinb_fwd              softmax, exp(library)*, NB/ZINB, sum(-1) modules.py:362,404; distributions.py:22-90
  elbo_reduce             means + weighting                       task.py:202,489
and the mirror-image backward kernels.
"""
from __future__ import annotations

import os

import ctypes as C
from typing import Callable, Dict, Optional

import torch

from . import _lib, ops
from .arena import ParamArena, _dec, _enc


def _tensors(view, cat, n_layers: int, zinb: bool) -> _lib.StepTensors:
    """Fill a vae_step_tensors from arena accessors (view(name) / cat(*names) -> tensor)."""
    t = _lib.StepTensors()
    ptr = lambda x: x.data_ptr()  # noqa: E731
    t.px_r = ptr(view("px_r"))
    t.enc_w0 = ptr(cat(_enc("z_encoder", 0, "w"), _enc("l_encoder", 0, "w")))
    t.enc_b0 = ptr(cat(_enc("z_encoder", 0, "b"), _enc("l_encoder", 0, "b")))
    for n in range(n_layers):
        t.encz_ln_g[n] = ptr(view(_enc("z_encoder", n, "g")))
        t.encz_ln_b[n] = ptr(view(_enc("z_encoder", n, "beta")))
        if n > 0:
            t.encz_w[n] = ptr(view(_enc("z_encoder", n, "w")))
            t.encz_b[n] = ptr(view(_enc("z_encoder", n, "b")))
        t.dec_w[n] = ptr(view(_dec(n, "w")))
        t.dec_b[n] = ptr(view(_dec(n, "b")))
        t.dec_ln_g[n] = ptr(view(_dec(n, "g")))
        t.dec_ln_b[n] = ptr(view(_dec(n, "beta")))
    t.encl_ln_g = ptr(view(_enc("l_encoder", 0, "g")))
    t.encl_ln_b = ptr(view(_enc("l_encoder", 0, "beta")))
    t.hz_w = ptr(cat("z_encoder.mean_encoder.weight", "z_encoder.var_encoder.weight"))
    t.hz_b = ptr(cat("z_encoder.mean_encoder.bias", "z_encoder.var_encoder.bias"))
    t.hl_w = ptr(cat("l_encoder.mean_encoder.weight", "l_encoder.var_encoder.weight"))
    t.hl_b = ptr(cat("l_encoder.mean_encoder.bias", "l_encoder.var_encoder.bias"))
    if zinb:
        t.heads_w = ptr(cat("decoder.px_scale_decoder.0.weight", "decoder.px_dropout_decoder.weight"))
        t.heads_b = ptr(cat("decoder.px_scale_decoder.0.bias", "decoder.px_dropout_decoder.bias"))
    else:
        t.heads_w = ptr(view("decoder.px_scale_decoder.0.weight"))
        t.heads_b = ptr(view("decoder.px_scale_decoder.0.bias"))
    return t


class CellHeads:
    """The cell-property heads on the native path (include/vae_amd.h, ``vae_cell_heads_step``): ``CellDecoder`` with
    LINEAR heads (modules.py:297-324; ``cell_decoder_hidden_layer=False`` is the shipped setting, config.py:67) and
    ``LitVAE._cell_loss`` (task.py:419-451) -- logits, masked cross-entropy (ignore_index -100) / squared error averaged
    over all cells, and the backward: d z (a ``stop_grad`` head contributes nothing, a gradient-reversal head ``-lambda``
    times its gradient) and the heads' weight / bias gradients, written into the arena.  Heads with a hidden layer stay
    on ``ElboStep.run(cell_head_fn=...)`` (torch autograd on the (B, L) latent)."""

    def __init__(self, arena: ParamArena, cell_properties: Dict[str, dict], grad_reverse: Optional[Dict[str, float]] = None,
                 prefix: str = "cell_decoder.cell_mlp.", class_weights: Optional[Dict[str, torch.Tensor]] = None):
        """``class_weights``: per discrete head, the ``weight`` of its ``nn.CrossEntropyLoss`` (``balance_classes``,
        task.py:129-132: ``clip(1 / freq, 0.1, 10)``); heads without an entry are unweighted."""
        self.lib = _lib.load()
        self.arena = arena
        names = list(cell_properties)
        if not 0 < len(names) <= _lib.VAE_MAX_HEADS:
            raise ValueError(f"CellHeads: 1..{_lib.VAE_MAX_HEADS} heads, got {len(names)}")
        h = _lib.CellHeads()
        h.n_heads = len(names)
        self.L = None
        self._class_weights = []
        for i, k in enumerate(names):
            wk, bk = f"{prefix}{k}.weight", f"{prefix}{k}.bias"
            if wk not in arena.offsets or bk not in arena.offsets:
                raise ValueError(f"CellHeads: head {k!r} is not a single Linear layer (hidden-layer heads: use cell_head_fn)")
            w = arena.p(wk)
            prop = cell_properties[k]
            width = len(prop["values"]) if prop["discrete"] else 1
            if w.dim() != 2 or w.shape[0] != width or (self.L is not None and w.shape[1] != self.L):
                raise ValueError(f"CellHeads: head {k!r}: weight shape {tuple(w.shape)} does not fit the property")
            self.L = w.shape[1]
            h.width[i], h.discrete[i] = width, int(bool(prop["discrete"]))
            if grad_reverse is not None and k in grad_reverse:          # CellDecoder._view_of_latent (modules.py:317-321)
                h.dz_coeff[i] = -float(grad_reverse[k])
            else:
                h.dz_coeff[i] = 0.0 if prop["stop_grad"] else 1.0
            h.w[i], h.b[i] = w.data_ptr(), arena.p(bk).data_ptr()
            h.dw[i], h.db[i] = arena.g(wk).data_ptr(), arena.g(bk).data_ptr()
            cw = (class_weights or {}).get(k)
            if cw is not None:
                if not prop["discrete"]:
                    raise ValueError(f"CellHeads: head {k!r} is continuous: class weights apply to discrete heads")
                cw = torch.as_tensor(cw, dtype=torch.float32).to(arena.device).contiguous()
                if cw.shape != (width,):
                    raise ValueError(f"CellHeads: head {k!r}: {width} class weights expected, got {tuple(cw.shape)}")
                self._class_weights.append(cw)     # keeps the device memory alive
                h.class_weight[i] = cw.data_ptr()
        if self.L > 64 or sum(h.width[i] for i in range(len(names))) > 64:
            raise ValueError("CellHeads: the native heads take a latent of at most 64 and at most 64 outputs in all")
        self.h, self.n = h, len(names)
        self._ws: Dict[int, torch.Tensor] = {}
        self._loss = torch.zeros(1, dtype=torch.float32, device=arena.device)

    def run(self, z: torch.Tensor, targets: torch.Tensor, mask: torch.Tensor, loss_scale: float = 1.0,
            dz: Optional[torch.Tensor] = None) -> torch.Tensor:
        """-> the heads' loss (0-dim GPU tensor, overwritten by the next call).  ``dz`` (B, L): receives
        d(loss_scale * loss)/dz and switches the weight / bias gradients on."""
        B = z.shape[0]
        for t, name in ((z, "z"), (targets, "targets"), (mask, "mask")):
            if not t.is_cuda or t.dtype != torch.float32 or t.dim() != 2 or t.stride(1) != 1 or t.shape[0] != B:
                raise ValueError(f"CellHeads: {name} must be a 2-D float32 GPU tensor with a contiguous last dim and B rows")
        if z.shape[1] != self.L or not z.is_contiguous() or targets.shape[1] < self.n or mask.shape[1] < self.n:
            raise ValueError("CellHeads: z must be contiguous (B, L); targets / mask need one column per head")
        if dz is not None and (dz.shape != z.shape or not dz.is_contiguous() or dz.dtype != torch.float32 or not dz.is_cuda):
            raise ValueError("CellHeads: dz must be a contiguous float32 GPU tensor shaped like z")
        if B not in self._ws:
            nbytes = self.lib.vae_cell_heads_workspace_bytes(B, self.L, C.byref(self.h))
            if nbytes <= 0:
                raise RuntimeError("vae_amd: bad cell-heads configuration")
            self._ws = {B: torch.empty(nbytes // 4 + 4, dtype=torch.float32, device=z.device)}
        ws = self._ws[B]
        rc = self.lib.vae_cell_heads_step(C.byref(self.h), z.data_ptr(), B, self.L, targets.data_ptr(), targets.stride(0),
                                          mask.data_ptr(), mask.stride(0), float(loss_scale),
                                          dz.data_ptr() if dz is not None else None, self._loss.data_ptr(), ws.data_ptr(),
                                          ws.numel() * 4, C.c_void_p(torch.cuda.current_stream().cuda_stream))
        _lib.check(rc, "cell_heads_step")
        return self._loss[0]


class ElboStep:
    """Fused fwd+bwd ELBO step over a ``ParamArena``.

    ``cell_head_fn(z) -> scalar loss`` (optional) is evaluated with torch autograd on the
    tiny (B, L) latent -- CellDecoder + _cell_loss (modules.py:312-324, task.py:419-451);
    its gradient w.r.t. z is added to the decoder's before the reparameterisation backward.
    """

    def __init__(self, arena: ParamArena, *, l_mean: float, l_var: float, kl_weight: float = 1.0,
                 gene_loss_coeff: float = 5e-3, precision: int = ops.F16X2, dropout_rate: float = 0.0, seed: int = 0,
                 input_dropout_rate: float = 0.0, input_dropout_alpha: float = 0.95,
                 gene_running: Optional[torch.Tensor] = None):
        """``precision``: ``ops.F16X2`` (default; fp32-grade: the five gene-sized products on two scaled f16 terms and
        three matrix-core products, everything else as ``ops.FP32``), ``ops.FP32`` (three bf16 terms, six products:
        element-wise fp32-grade for any operand range) or ``ops.BF16``.

        ``dropout_rate``: p of the nn.Dropout after the z-encoder / decoder LayerNorms (modules.py:115),
        applied by ``run(..., train=True)`` with masks from the library's Philox stream keyed by ``seed``
        (give every rank its own) and the step counter ``rng_step``.

        ``input_dropout_rate``: p of GeneDropout on the z-encoder's input (modules.py:190-209);
        ``gene_running`` (G,) is its ``running_non_zero_means`` state (created as zeros when omitted, updated in
        place by every training-mode run)."""
        if arena.likelihood not in ("zinb", "nb"):
            raise ValueError(f"ElboStep: likelihood must be 'zinb' or 'nb', got {arena.likelihood!r}")
        if not arena.flat.is_cuda:
            raise RuntimeError("ElboStep: the arena must live on the GPU (no CPU fallback)")
        self.lib = _lib.load()
        self.a = arena
        self.n_layers = arena.n_layers
        self.zinb = arena.likelihood == "zinb"
        self.l_mean, self.l_var = float(l_mean), float(l_var)
        self.kl_weight, self.coeff = float(kl_weight), float(gene_loss_coeff)
        self.precision = int(os.environ.get("VAE_STEP_PRECISION", precision))   # env: A/B the GEMM precision of a run
        if not 0.0 <= float(dropout_rate) < 1.0:
            raise ValueError(f"ElboStep: dropout_rate must be in [0, 1), got {dropout_rate}")
        self.dropout_rate, self.seed, self.rng_step = float(dropout_rate), int(seed) & (2 ** 64 - 1), 0
        if not 0.0 <= float(input_dropout_rate) <= 1.0:
            raise ValueError(f"ElboStep: input_dropout_rate must be in [0, 1], got {input_dropout_rate}")
        self.input_dropout_rate = float(input_dropout_rate) if input_dropout_rate >= 1e-6 else 0.0   # modules.py:199
        self.input_dropout_alpha = float(input_dropout_alpha)
        self.gene_running = gene_running
        a = arena
        self.H = a.p(_enc("z_encoder", 0, "w")).shape[0]
        self.Hl = a.p(_enc("l_encoder", 0, "w")).shape[0]
        self.G = a.p("px_r").shape[0]
        self.L = a.p("z_encoder.mean_encoder.weight").shape[0]
        self.Hd = a.p(_dec(0, "w")).shape[0]
        self.n_cov = int(getattr(a, "n_cov", 0))     # one-hot batch-covariate columns (modules.py:102-107); 0 = none
        if self.n_layers > _lib.VAE_MAX_LAYERS:
            raise ValueError(f"ElboStep: at most {_lib.VAE_MAX_LAYERS} layers")
        if self.input_dropout_rate > 0:
            if self.gene_running is None:
                self.gene_running = torch.zeros(self.G, dtype=torch.float32, device=a.device)
            r = self.gene_running
            if not r.is_cuda or r.dtype != torch.float32 or r.numel() != self.G or not r.is_contiguous():
                raise ValueError("ElboStep: gene_running must be a contiguous float32 GPU tensor of G elements")
        self.params = _tensors(a.p, a.pcat, self.n_layers, self.zinb)
        self.grads = _tensors(a.g, a.gcat, self.n_layers, self.zinb)
        self.weight_maxima = None
        self.weight_images = None
        self._maxima_source = self._images_source = None
        self._plans: Dict[int, tuple] = {}
        self.max_plans = 4          # distinct (batch size, training) workspaces kept alive (train, validation, two tails)
        self._ev_heads = self._ev_small = None
        self._ev_slabs = None
        self._zero = None
        self._last_fwd = None

    def use_weight_maxima(self, source) -> None:
        """Take the partial maxima of the two gene-sized weight matrices from ``source`` (a ``FusedAdamW`` built with
        ``track_weight_maxima=True``, or a (2, P) float32 GPU tensor kept current by the caller; ``None`` = scan again)
        instead of scanning 72 MB of weights at the start of every two-f16-term forward."""
        t = getattr(source, "weight_maxima", source)
        self._maxima_source = source if hasattr(source, "sync") else None
        if t is not None and (not torch.is_tensor(t) or t.dim() != 2 or t.shape[0] != 2 or t.dtype != torch.float32
                              or not t.is_cuda or not t.is_contiguous() or t.shape[1] % 4):
            raise ValueError("ElboStep.use_weight_maxima: need a contiguous (2, P) float32 GPU tensor, P a multiple of 4")
        self.weight_maxima = t

    def use_weight_images(self, source) -> None:
        """Take the bf16 images of the two gene-sized weight matrices from ``source`` (a ``FusedAdamW`` built with
        ``track_bf16_images=True``, or a pair of contiguous bfloat16 GPU tensors kept current by the caller: the first-layer
        encoder weights as stored in the arena, row padding included, and the decoder-head weights; ``None`` = cast again)
        instead of casting them at the start of every bf16-storage forward.  Other precision modes ignore them."""
        t = getattr(source, "weight_images", source)
        self._images_source = source if hasattr(source, "sync") else None
        if t is not None:
            want = ((self.H + self.Hl) * self.a.ld_enc_w, (2 if self.zinb else 1) * self.G * self.Hd)
            ok = isinstance(t, (tuple, list)) and len(t) == 2 and all(
                torch.is_tensor(q) and q.dtype == torch.bfloat16 and q.is_cuda and q.is_contiguous() and q.numel() == n
                for q, n in zip(t, want))
            if not ok:
                raise ValueError(f"ElboStep.use_weight_images: need two contiguous bfloat16 GPU tensors of {want} elements")
            t = tuple(t)
        self.weight_images = t

    def counts_path(self, B: int, train: bool = True) -> int:
        """How a step of B cells feeds float counts to the first-layer encoder products (include/vae_amd.h,
        vae_elbo_step_counts_path): 0 normalised matrix, 1 raw counts behind a statistics launch, 2 raw counts, optimistic."""
        return int(self.lib.vae_elbo_step_counts_path(C.byref(self._plan(B, 1.0, train)[0])))

    def bf16_activations(self, B: int, train: bool = False) -> bool:
        """Whether a step of batch size B keeps the logits / logit gradients as bf16 (include/vae_amd.h,
        vae_elbo_step_bf16_activations): the bf16 oracle is then the one with ``logits_bf16=True``."""
        return bool(self.lib.vae_elbo_step_bf16_activations(C.byref(self._plan(B, 1.0, train)[0])))

    def _plan(self, B: int, loss_scale: float, train: bool = False):
        """Config struct, workspace and named views for batch size B (cached).  GeneDropout changes the
        workspace layout, so training-mode plans with it are kept apart."""
        gene_drop = bool(train and self.input_dropout_rate > 0)
        key = (B, gene_drop)
        if key not in self._plans:
            cfg = _lib.StepConfig(B=B, G=self.G, H=self.H, Hl=self.Hl, L=self.L, Hd=self.Hd, n_layers=self.n_layers,
                                  zinb=int(self.zinb), precision=self.precision, l_mean=self.l_mean, l_var=self.l_var,
                                  kl_weight=self.kl_weight, gene_loss_coeff=self.coeff, loss_scale=1.0)
            if self.n_cov:
                cfg.n_cov, cfg.ld_cov, cfg.ld_enc_w = self.n_cov, self.n_cov, self.a.ld_enc_w
            if gene_drop:
                cfg.input_dropout_rate = self.input_dropout_rate
                cfg.input_dropout_alpha = self.input_dropout_alpha
                cfg.gene_running = self.gene_running.data_ptr()
            nbytes = self.lib.vae_elbo_step_workspace_bytes(C.byref(cfg))
            if nbytes <= 0:
                raise RuntimeError("vae_amd: bad step configuration")
            ws = torch.empty(nbytes // 4, dtype=torch.float32, device=self.a.device)
            off = {n: self.lib.vae_elbo_step_ws_offset(C.byref(cfg), i) for i, n in enumerate(_lib.WS_SLICES)}
            L = self.L
            v = dict(
                heads_z=ws[off["QZ_HEADS"]:off["QZ_HEADS"] + B * (L + 1)].view(B, L + 1),
                qz_v=ws[off["QZ_V"]:off["QZ_V"] + B].view(B, 1),
                z=ws[off["Z"]:off["Z"] + B * L].view(B, L),
                kl_z=ws[off["KL_Z"]:off["KL_Z"] + B],
                heads_l=ws[off["QL_HEADS"]:off["QL_HEADS"] + B * 2].view(B, 2),
                ql_v=ws[off["QL_V"]:off["QL_V"] + B].view(B, 1),
                library=ws[off["LIBRARY"]:off["LIBRARY"] + B].view(B, 1),
                kl_l=ws[off["KL_L"]:off["KL_L"] + B],
                ll=ws[off["LL"]:off["LL"] + B],
                scalars=ws[off["SCALARS"]:off["SCALARS"] + 5],
                dz_extra=ws[off["DZ_EXTRA"]:off["DZ_EXTRA"] + B * L].view(B, L),
                recon=ws[off["RECON"]:off["RECON"] + B],
            )
            if gene_drop:   # keep thresholds floor(p_g * 2^32) as int32 bit patterns, then mean_g(p_g)
                o = off["GENE_THRESHOLD"]
                v["gene_threshold"] = ws[o:o + self.G].view(torch.int32)
                v["gene_p_mean"] = ws[o + self.G:o + self.G + 1]
            self._plans[key] = (cfg, ws, nbytes, v)
            while len(self._plans) > self.max_plans:      # a workspace is 0.3-1.5 GB: keep the most recent batch sizes only
                self._plans.pop(next(iter(self._plans)))
        else:
            self._plans[key] = self._plans.pop(key)       # most recently used last
        cfg, ws, nbytes, v = self._plans[key]
        cfg.loss_scale = float(loss_scale)
        cfg.kl_weight = self.kl_weight
        for src in (self._maxima_source, self._images_source):   # stale after a load / broadcast? (ParamArena.touch)
            if src is not None:
                src.sync()
        wm = self.weight_maxima
        cfg.weight_max_enc = wm[0].data_ptr() if wm is not None else None
        cfg.weight_max_heads = wm[1].data_ptr() if wm is not None else None
        cfg.weight_max_n = wm.shape[1] if wm is not None else 0
        wi = self.weight_images
        cfg.weight_bf16_enc = wi[0].data_ptr() if wi is not None else None
        cfg.weight_bf16_heads = wi[1].data_ptr() if wi is not None else None
        cfg.x_dtype = 0
        return cfg, ws, nbytes, v

    # ------------------------------------------------------------------------------------
    @staticmethod
    def covariates(batch_labels: torch.Tensor, batch_mask: torch.Tensor, sizes) -> torch.Tensor:
        """The block FCLayers.forward appends to its input (modules.py:123-130): ``one_hot(label_p) * mask_p`` for every
        batch property p, (B, sum(sizes)) float32.  Negative labels are clamped to 0 IN PLACE, as the reference does."""
        batch_labels[batch_labels < 0] = 0
        cols = [torch.nn.functional.one_hot(batch_labels[:, n].long(), num_classes=int(k)) * batch_mask[:, n:n + 1]
                for n, k in enumerate(sizes)]
        return torch.cat(cols, dim=-1).to(torch.float32).contiguous()

    def run(self, x: torch.Tensor, eps_z: torch.Tensor, eps_l: torch.Tensor,
            cell_head_fn: Optional[Callable[[torch.Tensor], torch.Tensor]] = None,
            backward: bool = True, loss_scale: float = 1.0, reducer=None, train: Optional[bool] = None,
            rng_step: Optional[int] = None, batch_cov: Optional[torch.Tensor] = None,
            cell_heads: Optional[tuple] = None, fractional_counts: bool = False) -> Dict[str, torch.Tensor]:
        """One step.  The returned tensors are VIEWS of the step workspace: they are overwritten by
        the next ``run`` with the same batch size (clone what must survive).

        ``train`` (default: ``backward``) switches the dropout masks on; ``rng_step`` picks the mask set
        (default: a counter that advances with every training-mode run).

        ``cell_heads`` = ``(CellHeads, targets (B, P), mask (B, P))``: linear cell-property heads on the native path, instead
        of ``cell_head_fn``.

        ``fractional_counts``: the batch holds fractions (mixup / cutmix) -- a hint, never needed for correctness
        (``vae_step_config.fractional_counts``).

        ``batch_cov`` (B, n_cov): the masked one-hot batch covariates (``ElboStep.covariates``) of a model built with
        ``batch_properties``; required exactly then."""
        a = self.a
        if (batch_cov is not None) != (self.n_cov > 0):
            raise ValueError(f"ElboStep: the model has {self.n_cov} batch-covariate columns; batch_cov must be given exactly then")
        if batch_cov is not None and (not batch_cov.is_cuda or batch_cov.dtype != torch.float32 or not batch_cov.is_contiguous()
                                      or tuple(batch_cov.shape) != (x.shape[0], self.n_cov)):
            raise ValueError(f"ElboStep: batch_cov must be a contiguous float32 GPU tensor of shape {(x.shape[0], self.n_cov)}")
        for t, name in ((x, "x"), (eps_z, "eps_z"), (eps_l, "eps_l")):
            ok_dtype = t.dtype == torch.float32 or (name == "x" and t.dtype == torch.uint8)
            if not t.is_cuda or not ok_dtype or t.dim() != 2 or t.stride(1) != 1:
                raise ValueError(f"ElboStep: {name} must be a 2-D float32 (x: or uint8) GPU tensor with a contiguous last dim")
        B, G = x.shape
        L = self.L
        if G != self.G:
            raise ValueError(f"ElboStep: x has {G} genes, model has {self.G}")
        if eps_z.shape != (B, L) or eps_l.shape != (B, 1) or not eps_z.is_contiguous() or not eps_l.is_contiguous():
            raise ValueError("ElboStep: eps_z must be contiguous (B, L) and eps_l (B, 1)")
        train = backward if train is None else bool(train)
        cfg, ws, nbytes, v = self._plan(B, loss_scale, train)
        cfg.x_dtype = 1 if x.dtype == torch.uint8 else 0   # uint8 counts: the reference's on-disk format (f2)
        cfg.batch_cov = batch_cov.data_ptr() if batch_cov is not None else None
        cfg.dropout_rate = self.dropout_rate if train else 0.0
        cfg.rng_seed = self.seed
        if rng_step is None:
            rng_step = self.rng_step
            self.rng_step += int(train and (self.dropout_rate > 0 or self.input_dropout_rate > 0))
        cfg.rng_step = int(rng_step) & 0xFFFFFFFF
        # a training step: the ELBO reduction (recon, the scalars) rides in the backward's first launch (vae_step_config)
        cfg.defer_elbo_reduce = int(bool(backward))
        cfg.fractional_counts = int(bool(fractional_counts))
        stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        ldx = x.stride(0) if B > 1 else G
        rc = self.lib.vae_elbo_step_forward(C.byref(cfg), C.byref(self.params), x.data_ptr(), ldx, eps_z.data_ptr(),
                                            eps_l.data_ptr(), ws.data_ptr(), nbytes, stream)
        _lib.check(rc, "elbo_step_forward")
        s = v["scalars"]
        out = dict(qz_m=v["heads_z"][:, :L], qz_v=v["qz_v"], z=v["z"], ql_m=v["heads_l"][:, :1], ql_v=v["ql_v"],
                   library=v["library"], recon=v["recon"], kl_z=v["kl_z"], kl_l=v["kl_l"], recon_mean=s[0],
                   gene_loss=s[3], scalars=s)
        if "gene_threshold" in v:
            out["gene_threshold"], out["gene_p_mean"] = v["gene_threshold"], v["gene_p_mean"]

        # ---------------- cell heads (tiny; torch autograd on z only) ----------------
        use_extra = 0
        if cell_heads is not None:
            if cell_head_fn is not None:
                raise ValueError("ElboStep: give cell_heads or cell_head_fn, not both")
            heads, t_heads, m_heads = cell_heads
            out["cell_loss"] = heads.run(v["z"], t_heads, m_heads, loss_scale, v["dz_extra"] if backward else None)
            use_extra = int(backward)
        elif cell_head_fn is not None:
            z_leaf = v["z"].detach().clone().requires_grad_(True)
            with torch.enable_grad():
                cell_loss = cell_head_fn(z_leaf)
            if backward:
                a.zero_cell_grads()   # autograd accumulates into the arena views; every other gradient is overwritten
                (cell_loss * loss_scale).backward()
                v["dz_extra"].copy_(z_leaf.grad)
                use_extra = 1
            out["cell_loss"] = cell_loss.detach()
        else:
            if self._zero is None:
                self._zero = torch.zeros((), device=x.device)
            out["cell_loss"] = self._zero
            out["loss"] = s[4]
        if not backward:
            if "loss" not in out:
                out["loss"] = s[4] + out["cell_loss"]
            self._last_fwd = (B, cfg, ws, nbytes)
            return out
        self._last_fwd = None   # the backward overwrites the logits in place

        # Data parallel: the native backward records two HIP events on the compute stream -- decoder-head weight
        # gradients final / everything but the first-layer encoder weights final -- and the reducer's side stream
        # waits for them, so the buckets travel while the rest of the backward computes (parallel.py).
        ev_heads = ev_small = None
        if reducer is not None and getattr(reducer, "active", reducer.world > 1) and reducer.on_gpu and not os.environ.get("VAE_NO_NATIVE_EVENT"):
            if self._ev_heads is None:
                self._ev_heads, self._ev_small = torch.cuda.Event(), torch.cuda.Event()
                self._ev_heads.record()   # materialises the hipEvent_t handles (torch creates them lazily)
                self._ev_small.record()
            ev_heads, ev_small = self._ev_heads, self._ev_small
        n_slabs = int(getattr(reducer, "enc_slabs", 1)) if ev_heads is not None else 1
        if n_slabs > 1:   # the encoder weight gradient in row slabs, each released by its own event (parallel.py)
            if self._ev_slabs is None or len(self._ev_slabs) != n_slabs:
                self._ev_slabs = [torch.cuda.Event() for _ in range(n_slabs)]
                for e in self._ev_slabs:
                    e.record()
            bounds = (C.c_int64 * (n_slabs + 1))(*reducer.enc_row_bounds)
            evs = (C.c_void_p * n_slabs)(*[e.cuda_event for e in self._ev_slabs])
            rc = self.lib.vae_elbo_step_backward_slabs(
                C.byref(cfg), C.byref(self.params), C.byref(self.grads), x.data_ptr(), ldx, eps_z.data_ptr(), eps_l.data_ptr(),
                use_extra, ws.data_ptr(), nbytes, C.c_void_p(ev_heads.cuda_event), C.c_void_p(ev_small.cuda_event), bounds, evs,
                n_slabs, stream)
        else:
            rc = self.lib.vae_elbo_step_backward_events(
                C.byref(cfg), C.byref(self.params), C.byref(self.grads), x.data_ptr(), ldx, eps_z.data_ptr(), eps_l.data_ptr(),
                use_extra, ws.data_ptr(), nbytes, C.c_void_p(ev_heads.cuda_event) if ev_heads is not None else None,
                C.c_void_p(ev_small.cuda_event) if ev_small is not None else None, stream)
        _lib.check(rc, "elbo_step_backward")
        if "loss" not in out:   # (after the backward's first launch: that is where a training step's scalars are written)
            out["loss"] = s[4] + out["cell_loss"]
        if reducer is not None:
            reducer.bucket_ready("dec_heads", after_event=ev_heads)
            reducer.bucket_ready("small", after_event=ev_small)
            if n_slabs > 1:
                for i, e in enumerate(self._ev_slabs):
                    reducer.bucket_ready(f"enc_w0:{i}", after_event=e)
            else:
                for tag in [t for t in reducer.buckets if t.startswith("enc_w0")]:   # (slabs without native events: all at the end)
                    reducer.bucket_ready(tag)
        return out

    # ------------------------------------------------------------------------------------
    def px_rate(self, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """The decoder mean px_rate = exp(library) * softmax(logits), (B, G), of the LAST forward-only
        ``run(..., backward=False)`` -- what the validation step exports (task.py:240,263)."""
        if self._last_fwd is None:
            raise RuntimeError("ElboStep.px_rate: call run(..., backward=False) first (a backward overwrites the logits)")
        B, cfg, ws, nbytes = self._last_fwd
        if out is None:
            out = torch.empty(B, self.G, dtype=torch.float32, device=ws.device)
        if out.shape != (B, self.G) or out.dtype != torch.float32 or not out.is_cuda or out.stride(1) != 1:
            raise ValueError(f"ElboStep.px_rate: out must be a float32 GPU tensor of shape {(B, self.G)}")
        stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        rc = self.lib.vae_elbo_step_px_rate(C.byref(cfg), ws.data_ptr(), nbytes, out.data_ptr(), out.stride(0), stream)
        _lib.check(rc, "elbo_step_px_rate")
        return out

    def infer(self, x: torch.Tensor, eps_z: torch.Tensor, eps_l: torch.Tensor, *, want_px_rate: bool = False,
              cell_head_fn: Optional[Callable[[torch.Tensor], torch.Tensor]] = None,
              batch_cov: Optional[torch.Tensor] = None) -> Dict[str, torch.Tensor]:
        """Validation / inference pass (LitVAE.validation_step, task.py:235-282): the forward alone, no
        gradient is touched; ``px_rate`` (B, G) on request."""
        out = self.run(x, eps_z, eps_l, cell_head_fn=cell_head_fn, backward=False, batch_cov=batch_cov)
        if want_px_rate:
            out["px_rate"] = self.px_rate()
        return out
