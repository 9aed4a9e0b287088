"""NB / ZINB likelihoods with the reference's API (src/distributions.py), computed by the
element-wise HIP kernels of csrc/likelihood.hip.

``log_zinb_positive(x, mu, theta, pi, eps=1e-8)`` (distributions.py:22-62) and
``log_nb_positive(x, mu, theta, eps=1e-8)`` (:65-90) take (B, G) tensors, theta either (G,)
or broadcastable to (B, G), and return the (B, G) log-likelihood; gradients flow to mu,
theta and pi.  The kernels hard-code eps = 1e-8, the only value the reference uses.
"""
from __future__ import annotations

import warnings
from typing import Tuple, Union

import torch
from torch.distributions import Distribution, Gamma, constraints
from torch.distributions import Poisson as _TorchPoisson
from torch.distributions.utils import broadcast_all, lazy_property, logits_to_probs, probs_to_logits

from .autograd import LogLikFn, PoissonLogProbFn


def _check_eps(eps: float) -> None:
    if abs(eps - 1e-8) > 1e-12:
        raise ValueError("vae_amd.distributions: the HIP kernels implement eps = 1e-8 only")


def _as2d(t: torch.Tensor, like: torch.Tensor) -> torch.Tensor:
    if t.dim() == 1:
        return t
    return t.expand_as(like) if t.shape != like.shape else t


def log_zinb_positive(x, mu, theta, pi, eps=1e-8):
    _check_eps(eps)
    mu, pi = mu.expand_as(x), pi.expand_as(x)
    th = theta
    if th.dim() == 2 and th.shape[0] != 1 and th.shape != x.shape:
        th = th.expand_as(x)
    return LogLikFn.apply(x, mu, th, pi, True)


def log_nb_positive(x, mu, theta, eps=1e-8):
    _check_eps(eps)
    mu = mu.expand_as(x)
    th = theta
    if th.dim() == 2 and th.shape[0] != 1 and th.shape != x.shape:
        th = th.expand_as(x)
    return LogLikFn.apply(x, mu, th, None, False)


class Poisson(_TorchPoisson):
    """torch's Poisson (what the reference imports, task.py:14,168) with ``log_prob`` on the HIP kernel for the
    2-D float32 GPU case of the reconstruction loss; anything else takes torch's own expression."""

    def log_prob(self, value):
        r = self.rate
        if torch.is_tensor(value) and value.is_cuda and value.dim() == 2 and value.dtype == torch.float32 and \
                r.dtype == torch.float32 and r.shape == value.shape:
            if self._validate_args:
                self._validate_sample(value)
            return PoissonLogProbFn.apply(value, r)
        return super().log_prob(value)


def _convert_mean_disp_to_counts_logits(mu, theta, eps=1e-6):
    assert (mu is None) == (theta is None), "If using the mu/theta NB parameterization, both parameters must be specified"
    return theta, (mu + eps).log() - (theta + eps).log()


def _convert_counts_logits_to_mean_disp(total_count, logits):
    return logits.exp() * total_count, total_count


class NegativeBinomial(Distribution):
    """NB in the (mu, theta) or (total_count, probs/logits) parameterisation (distributions.py:121-196)."""

    arg_constraints = {"mu": constraints.greater_than_eq(0), "theta": constraints.greater_than_eq(0)}
    support = constraints.nonnegative_integer

    def __init__(self, total_count=None, probs=None, logits=None, mu=None, theta=None, validate_args=True):
        self._eps = 1e-8
        if (mu is None) == (total_count is None):
            raise ValueError(
                "Please use one of the two possible parameterizations. Refer to the documentation for more information.")
        if total_count is not None and (logits is not None or probs is not None):
            logits = logits if logits is not None else probs_to_logits(probs)
            total_count = total_count.type_as(logits)
            total_count, logits = broadcast_all(total_count, logits)
            mu, theta = _convert_counts_logits_to_mean_disp(total_count, logits)
            self._theta_vec = None
        else:
            self._theta_vec = theta if theta.dim() == 1 else None  # keep the (G,) form: no B-fold expansion
            mu, theta = broadcast_all(mu, theta)
        self.mu, self.theta = mu, theta
        super().__init__(validate_args=validate_args)

    def _theta_arg(self):
        return self._theta_vec if self._theta_vec is not None else self.theta

    def sample(self, sample_shape=torch.Size()):
        p_means = self._gamma().sample(sample_shape)
        return _TorchPoisson(torch.clamp(p_means, max=1e8)).sample()

    def _warn_support(self, value):
        try:
            self._validate_sample(value)
        except ValueError:
            warnings.warn("The value argument must be within the support of the distribution", UserWarning)

    def log_prob(self, value):
        if self._validate_args:
            self._warn_support(value)
        return log_nb_positive(value, mu=self.mu, theta=self._theta_arg(), eps=self._eps)

    def _gamma(self):
        return Gamma(concentration=self.theta, rate=self.theta / self.mu)


class ZeroInflatedNegativeBinomial(NegativeBinomial):
    """ZINB with zero-inflation logits (distributions.py:199-270)."""

    arg_constraints = {
        "mu": constraints.greater_than_eq(0),
        "theta": constraints.greater_than_eq(0),
        "zi_probs": constraints.half_open_interval(0.0, 1.0),
        "zi_logits": constraints.real,
    }
    support = constraints.nonnegative_integer

    def __init__(self, total_count=None, probs=None, logits=None, mu=None, theta=None, zi_logits=None,
                 validate_args=True):
        super().__init__(total_count=total_count, probs=probs, logits=logits, mu=mu, theta=theta,
                         validate_args=validate_args)
        self.zi_logits, self.mu, self.theta = broadcast_all(zi_logits, self.mu, self.theta)

    @lazy_property
    def zi_logits(self) -> torch.Tensor:
        return probs_to_logits(self.zi_probs, is_binary=True)

    @lazy_property
    def zi_probs(self) -> torch.Tensor:
        return logits_to_probs(self.zi_logits, is_binary=True)

    def sample(self, sample_shape: Union[torch.Size, Tuple] = torch.Size()) -> torch.Tensor:
        with torch.no_grad():
            samp = super().sample(sample_shape=sample_shape)
            samp[torch.rand_like(samp) <= self.zi_probs] = 0.0
            return samp

    def log_prob(self, value: torch.Tensor) -> torch.Tensor:
        self._warn_support(value)
        return log_zinb_positive(value, self.mu, self._theta_arg(), self.zi_logits, eps=1e-08)
