"""The fused VAE training step (forward + backward ELBO) on hand-written gfx950 kernels.

This is the hot path of the reference's ``LitVAE.training_step`` (task.py:453-502):
z/l encoders -> decoder -> exp(px_r) -> Gaussian KLs + NB/ZINB reconstruction ->
``loss = gene_loss_coeff * gene_loss + cell_loss`` -> gradients of every parameter.
It runs as a fixed sequence of kernel launches on the current HIP stream -- no autograd
graph, no (B,G) temporaries beyond the logits buffer, which is overwritten in place by its
own gradient.  Parameters and gradients live in a ``ParamArena``.

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

from typing import Callable, Dict, Optional

import torch

from . import ops
from .arena import ParamArena, _dec, _enc


class ElboStep:
    """Fused fwd+bwd ELBO step over a ``ParamArena``.

    ``cell_head_fn(z) -> scalar loss`` (optional) is evaluated with torch autograd on the
    tiny (B, L) latent -- CellDecoder + _cell_loss (modules.py:312-324, task.py:419-451);
    its gradient w.r.t. z is added to the decoder's before the reparameterisation backward.
    """

    def __init__(self, arena: ParamArena, *, l_mean: float, l_var: float, kl_weight: float = 1.0,
                 gene_loss_coeff: float = 5e-3, precision: int = ops.FP32):
        self.a = arena
        self.n_layers = arena.n_layers
        self.zinb = arena.likelihood == "zinb"
        if arena.likelihood not in ("zinb", "nb"):
            raise ValueError(f"ElboStep: likelihood must be 'zinb' or 'nb', got {arena.likelihood!r}")
        self.l_mean, self.l_var = float(l_mean), float(l_var)
        self.kl_weight, self.coeff = float(kl_weight), float(gene_loss_coeff)
        self.precision = precision
        a = arena
        self.H = a.p(_enc("z_encoder", 0, "w")).shape[0]
        self.Hl = a.p(_enc("l_encoder", 0, "w")).shape[0]
        self.G = a.p("px_r").shape[0]
        self.L = a.p("z_encoder.mean_encoder.weight").shape[0]
        self.Hd = a.p(_dec(0, "w")).shape[0]
        self.table = ops.gene_table(a.p("px_r")) if a.flat.is_cuda else None

    # ------------------------------------------------------------------------------------
    def run(self, x: torch.Tensor, eps_z: torch.Tensor, eps_l: torch.Tensor,
            cell_head_fn: Optional[Callable[[torch.Tensor], torch.Tensor]] = None,
            backward: bool = True, loss_scale: float = 1.0, reducer=None) -> Dict[str, torch.Tensor]:
        a, P = self.a, self.precision
        B, G = x.shape
        H, Hl, L, Hd, nl = self.H, self.Hl, self.L, self.Hd, self.n_layers
        if G != self.G:
            raise ValueError(f"ElboStep: x has {G} genes, model has {self.G}")
        if eps_z.shape != (B, L) or eps_l.shape != (B, 1):
            raise ValueError("ElboStep: eps_z must be (B, L) and eps_l (B, 1)")
        gemm = ops.gemm

        # ---------------- forward ----------------
        mean = ops.row_mean(x)
        w_enc0 = a.pcat(_enc("z_encoder", 0, "w"), _enc("l_encoder", 0, "w"))
        b_enc0 = a.pcat(_enc("z_encoder", 0, "b"), _enc("l_encoder", 0, "b"))
        pre_enc0 = gemm(ops.NT, x, w_enc0, bias=b_enc0, x_mean=mean, precision=P)          # (B, H+Hl)
        h_enc0 = torch.empty(B, H + Hl, dtype=torch.float32, device=x.device)
        _, mz0, rz0 = ops.gelu_ln_fwd(pre_enc0[:, :H], a.p(_enc("z_encoder", 0, "g")), a.p(_enc("z_encoder", 0, "beta")),
                                      out=h_enc0[:, :H])
        _, ml, rl = ops.gelu_ln_fwd(pre_enc0[:, H:], a.p(_enc("l_encoder", 0, "g")), a.p(_enc("l_encoder", 0, "beta")),
                                    out=h_enc0[:, H:])
        hz, hl = h_enc0[:, :H], h_enc0[:, H:]
        z_layers = [(None, pre_enc0[:, :H], mz0, rz0)]  # (input, pre, mean, rstd) per z-encoder layer
        for n in range(1, nl):
            pre = gemm(ops.NT, hz, a.p(_enc("z_encoder", n, "w")), bias=a.p(_enc("z_encoder", n, "b")), precision=P)
            hz_n, m_, r_ = ops.gelu_ln_fwd(pre, a.p(_enc("z_encoder", n, "g")), a.p(_enc("z_encoder", n, "beta")))
            z_layers.append((hz, pre, m_, r_))
            hz = hz_n

        w_hz = a.pcat("z_encoder.mean_encoder.weight", "z_encoder.var_encoder.weight")     # (L+1, H)
        b_hz = a.pcat("z_encoder.mean_encoder.bias", "z_encoder.var_encoder.bias")
        w_hl = a.pcat("l_encoder.mean_encoder.weight", "l_encoder.var_encoder.weight")     # (2, Hl)
        b_hl = a.pcat("l_encoder.mean_encoder.bias", "l_encoder.var_encoder.bias")
        heads_z = gemm(ops.NT, hz, w_hz, bias=b_hz, precision=P)
        heads_l = gemm(ops.NT, hl, w_hl, bias=b_hl, precision=P)
        qz_v, z, kl_z = ops.reparam_kl_fwd(heads_z, eps_z, 0.0, 1.0)
        ql_v, library, kl_l = ops.reparam_kl_fwd(heads_l, eps_l, self.l_mean, self.l_var)

        d_layers = []
        px = z
        for n in range(nl):
            pre = gemm(ops.NT, px, a.p(_dec(n, "w")), bias=a.p(_dec(n, "b")), precision=P)
            px_n, m_, r_ = ops.gelu_ln_fwd(pre, a.p(_dec(n, "g")), a.p(_dec(n, "beta")))
            d_layers.append((px, pre, m_, r_))
            px = px_n

        if self.zinb:
            w_heads = a.pcat("decoder.px_scale_decoder.0.weight", "decoder.px_dropout_decoder.weight")  # (2G, Hd)
            b_heads = a.pcat("decoder.px_scale_decoder.0.bias", "decoder.px_dropout_decoder.bias")
        else:
            w_heads, b_heads = a.p("decoder.px_scale_decoder.0.weight"), a.p("decoder.px_scale_decoder.0.bias")
        logits = gemm(ops.NT, px, w_heads, bias=b_heads, precision=P)                        # (B, 2G) or (B, G)
        ls = logits[:, :G]
        lp = logits[:, G:] if self.zinb else None

        ops.gene_table(a.p("px_r"), out=self.table)
        lib1 = library.view(B)
        ll, rowmax, rowsum, S = ops.nbzinb_fwd(self.zinb, ls, lp, x, lib1, self.table)
        scalars = ops.elbo_reduce(ll, kl_z, kl_l, self.kl_weight, self.coeff)

        out = dict(qz_m=heads_z[:, :L], qz_v=qz_v.view(B, 1), z=z, ql_m=heads_l[:, :1], ql_v=ql_v.view(B, 1),
                   library=library, recon=-ll, kl_z=kl_z, kl_l=kl_l, recon_mean=scalars[0], gene_loss=scalars[3],
                   scalars=scalars)

        # ---------------- cell heads (tiny; torch autograd on z only) ----------------
        dz_extra = None
        cell_loss = None
        if cell_head_fn is not None:
            z_leaf = z.detach().requires_grad_(True)
            with torch.enable_grad():
                cell_loss = cell_head_fn(z_leaf)
            if backward:
                a.zero_small_grads()
                (cell_loss * loss_scale).backward()
                dz_extra = z_leaf.grad
            cell_loss = cell_loss.detach()
            out["cell_loss"] = cell_loss
            out["loss"] = scalars[4] + cell_loss
        else:
            out["cell_loss"] = torch.zeros((), device=x.device)
            out["loss"] = scalars[4]
            if backward:
                a.zero_small_grads()
        if not backward:
            return out

        # ---------------- backward ----------------
        # d loss / d ll[b] = -coeff/B ; d loss / d kl[b] = coeff*kl_weight/B   (task.py:202,489)
        g_ll = -self.coeff / B * loss_scale
        g_kl = self.coeff * self.kl_weight / B * loss_scale
        dlibrary = torch.empty(B, dtype=torch.float32, device=x.device)
        if self.zinb:
            db_s, db_p = a.g("decoder.px_scale_decoder.0.bias"), a.g("decoder.px_dropout_decoder.bias")
        else:
            db_s, db_p = a.g("decoder.px_scale_decoder.0.bias"), None
        ops.nbzinb_bwd(self.zinb, ls, lp, x, lib1, self.table, rowmax, rowsum, S, None, g_ll, ls, lp, dlibrary,
                       a.g("px_r"), db_s, db_p)                                             # logits := dlogits
        if self.zinb:
            gw_heads = a.gcat("decoder.px_scale_decoder.0.weight", "decoder.px_dropout_decoder.weight")
        else:
            gw_heads = a.g("decoder.px_scale_decoder.0.weight")
        gemm(ops.TN, logits, px, out=gw_heads, precision=P)                                  # dW = dlogits^T px
        if reducer is not None:
            reducer.bucket_ready("dec_heads")  # the largest gradient travels while the rest computes
        dpx = gemm(ops.NN, logits, w_heads, precision=P)                                     # (B, Hd)
        del logits, ls, lp

        for n in reversed(range(nl)):
            inp, pre, m_, r_ = d_layers[n]
            dpre = ops.gelu_ln_bwd(dpx, pre, m_, r_, a.p(_dec(n, "g")), a.g(_dec(n, "g")), a.g(_dec(n, "beta")),
                                   a.g(_dec(n, "b")), dpre=dpx)
            gemm(ops.TN, dpre, inp, out=a.g(_dec(n, "w")), precision=P)
            dpx = gemm(ops.NN, dpre, a.p(_dec(n, "w")), precision=P)
        dz = dpx                                                                             # (B, L)
        if dz_extra is not None:
            dz = dz + dz_extra

        dheads_z = ops.reparam_kl_bwd(dz, None, g_kl, heads_z, eps_z, qz_v, 0.0, 1.0)
        dheads_l = ops.reparam_kl_bwd(dlibrary.view(B, 1), None, g_kl, heads_l, eps_l, ql_v, self.l_mean, self.l_var)
        gemm(ops.TN, dheads_z, hz, out=a.gcat("z_encoder.mean_encoder.weight", "z_encoder.var_encoder.weight"), precision=P)
        gemm(ops.TN, dheads_l, hl, out=a.gcat("l_encoder.mean_encoder.weight", "l_encoder.var_encoder.weight"), precision=P)
        ops.colsum_into(dheads_z, a.gcat("z_encoder.mean_encoder.bias", "z_encoder.var_encoder.bias"))
        ops.colsum_into(dheads_l, a.gcat("l_encoder.mean_encoder.bias", "l_encoder.var_encoder.bias"))
        dh = gemm(ops.NN, dheads_z, w_hz, precision=P)                                       # (B, H)
        dpre_enc0 = torch.empty(B, H + Hl, dtype=torch.float32, device=x.device)
        gemm(ops.NN, dheads_l, w_hl, out=dpre_enc0[:, H:], precision=P)                      # dhl, turned into dpre below

        for n in reversed(range(1, nl)):
            inp, pre, m_, r_ = z_layers[n]
            dpre = ops.gelu_ln_bwd(dh, pre, m_, r_, a.p(_enc("z_encoder", n, "g")), a.g(_enc("z_encoder", n, "g")),
                                   a.g(_enc("z_encoder", n, "beta")), a.g(_enc("z_encoder", n, "b")), dpre=dh)
            gemm(ops.TN, dpre, inp, out=a.g(_enc("z_encoder", n, "w")), precision=P)
            dh = gemm(ops.NN, dpre, a.p(_enc("z_encoder", n, "w")), precision=P)
        ops.gelu_ln_bwd(dh, pre_enc0[:, :H], mz0, rz0, a.p(_enc("z_encoder", 0, "g")), a.g(_enc("z_encoder", 0, "g")),
                        a.g(_enc("z_encoder", 0, "beta")), a.g(_enc("z_encoder", 0, "b")), dpre=dpre_enc0[:, :H])
        ops.gelu_ln_bwd(dpre_enc0[:, H:], pre_enc0[:, H:], ml, rl, a.p(_enc("l_encoder", 0, "g")),
                        a.g(_enc("l_encoder", 0, "g")), a.g(_enc("l_encoder", 0, "beta")), a.g(_enc("l_encoder", 0, "b")),
                        dpre=dpre_enc0[:, H:])
        gemm(ops.TN, dpre_enc0, x, out=a.gcat(_enc("z_encoder", 0, "w"), _enc("l_encoder", 0, "w")), x_mean=mean,
             precision=P)                                                                    # dW = dpre^T (x/mean-1)
        if reducer is not None:
            reducer.bucket_ready("rest")
        return out
