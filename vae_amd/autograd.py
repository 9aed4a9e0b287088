"""torch.autograd.Function wrappers: each fused stage is one call into the C-ABI HIP library
(forward) and one or more calls (backward).  These give the reference's nn.Module API
(vae_amd/modules.py, networks.py, distributions.py) a HIP implementation op by op; the
training fast path (vae_amd/step.py) bypasses autograd entirely."""
from __future__ import annotations

import torch
from torch.autograd import Function

from . import ops


def _c(t: torch.Tensor) -> torch.Tensor:
    return t if (t.dim() == 2 and t.stride(1) == 1 and t.dtype == torch.float32) else t.contiguous().float()


class LinearFn(Function):
    """y = x W^T + b (nn.Linear) on the MFMA GEMM; backward = dgrad (NN) + wgrad (TN) + column sum."""

    @staticmethod
    def forward(ctx, x, w, b, precision):
        x, w = _c(x), _c(w)
        ctx.save_for_backward(x, w)
        ctx.has_bias = b is not None
        ctx.precision = precision
        return ops.gemm(ops.NT, x, w, bias=b, precision=precision)

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        dy = _c(dy)
        dx = ops.gemm(ops.NN, dy, w, precision=ctx.precision) if ctx.needs_input_grad[0] else None
        dw = ops.gemm(ops.TN, dy, x, precision=ctx.precision) if ctx.needs_input_grad[1] else None
        db = None
        if ctx.has_bias and ctx.needs_input_grad[2]:
            db = ops.colsum_into(dy, torch.zeros(dy.shape[1], dtype=torch.float32, device=dy.device))
        return dx, dw, db, None


class CountLinearFn(Function):
    """y = (x / mean_g(x) - 1) W^T + b: the encoders' first layer on RAW counts
    (Encoder.forward, modules.py:264,267).  The normalised matrix is never materialised:
    the GEMM normalises the count operand while staging it.  x is data: no gradient."""

    @staticmethod
    def forward(ctx, x, w, b, precision):
        x, w = _c(x), _c(w)
        mean = ops.row_mean(x)
        ctx.save_for_backward(x, mean)
        ctx.has_bias = b is not None
        ctx.precision = precision
        return ops.gemm(ops.NT, x, w, bias=b, x_mean=mean, precision=precision)

    @staticmethod
    def backward(ctx, dy):
        x, mean = ctx.saved_tensors
        dy = _c(dy)
        if ctx.needs_input_grad[0]:
            raise RuntimeError("vae_amd: gradients w.r.t. the count matrix are not part of the training path")
        dw = ops.gemm(ops.TN, dy, x, x_mean=mean, precision=ctx.precision) if ctx.needs_input_grad[1] else None
        db = None
        if ctx.has_bias and ctx.needs_input_grad[2]:
            db = ops.colsum_into(dy, torch.zeros(dy.shape[1], dtype=torch.float32, device=dy.device))
        return None, dw, db, None


class GeluLayerNormFn(Function):
    """LayerNorm(GELU(pre)) with affine parameters, eps 1e-5 (FCLayers, modules.py:112-114)."""

    @staticmethod
    def forward(ctx, pre, gamma, beta):
        pre = _c(pre)
        out, mean, rstd = ops.gelu_ln_fwd(pre, gamma.contiguous(), beta.contiguous())
        ctx.save_for_backward(pre, mean, rstd, gamma)
        return out

    @staticmethod
    def backward(ctx, dout):
        pre, mean, rstd, gamma = ctx.saved_tensors
        W = pre.shape[1]
        dg = torch.zeros(W, dtype=torch.float32, device=pre.device)
        db = torch.zeros(W, dtype=torch.float32, device=pre.device)
        dpre = ops.gelu_ln_bwd(_c(dout), pre, mean, rstd, gamma.contiguous(), dg, db, None)
        return dpre, dg, db


class SoftmaxRateFn(Function):
    """(px_scale, px_rate) = (softmax(logits), exp(library) * softmax(logits)); modules.py:401,404."""

    @staticmethod
    def forward(ctx, logits, library):
        logits = _c(logits)
        lib1 = library.reshape(-1).contiguous()
        ps, pr = ops.softmax_rate_fwd(logits, lib1)
        ctx.save_for_backward(ps, lib1)
        ctx.lib_shape = library.shape
        return ps, pr

    @staticmethod
    def backward(ctx, d_scale, d_rate):
        ps, lib1 = ctx.saved_tensors
        d_scale = None if d_scale is None else d_scale.contiguous()
        d_rate = None if d_rate is None else d_rate.contiguous()
        dl, dlib = ops.softmax_rate_bwd(ps, lib1, d_scale, d_rate)
        return dl, dlib.view(ctx.lib_shape)


class PoissonLogProbFn(Function):
    """Poisson(rate).log_prob(x) (task.py:167-168) on the HIP kernels; gradient to the rate only."""

    @staticmethod
    def forward(ctx, x, rate):
        x, rate = _c(x), _c(rate)
        ctx.save_for_backward(x, rate)
        return ops.poisson_fwd(x, rate)

    @staticmethod
    def backward(ctx, gout):
        x, rate = ctx.saved_tensors
        return None, ops.poisson_bwd(_c(gout), x, rate)


class LogLikFn(Function):
    """Element-wise log_nb_positive / log_zinb_positive (distributions.py:22-90)."""

    @staticmethod
    def forward(ctx, x, mu, theta, pi, zinb):
        x, mu = _c(x), _c(mu)
        theta = theta.contiguous()
        pi = _c(pi) if zinb else None
        ctx.zinb = zinb
        ctx.theta_shape = theta.shape
        ctx.save_for_backward(x, mu, theta, pi)
        return ops.loglik_elem_fwd(zinb, x, mu, theta, pi)

    @staticmethod
    def backward(ctx, gout):
        x, mu, theta, pi = ctx.saved_tensors
        dmu, dth, dpi = ops.loglik_elem_bwd(ctx.zinb, _c(gout), x, mu, theta, pi)
        if theta.dim() == 1 or (theta.dim() == 2 and theta.shape[0] == 1 and x.shape[0] > 1):
            red = ops.colsum_into(dth, torch.zeros(dth.shape[1], dtype=torch.float32, device=dth.device))
            dth = red.view(ctx.theta_shape)
        return None, dmu, dth, dpi, None
