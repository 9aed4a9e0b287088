"""Gradient reversal for the adversarial cell heads (role of the reference's src/functional.py:8-30):
the forward pass is the identity, the backward pass multiplies the incoming gradient by -lambd.
Host-side plumbing of the module API; the fused step applies the same sign flip through dz_extra."""
from typing import Optional, Tuple

import torch
from torch import nn


class GradReverse(torch.autograd.Function):
    """``GradReverse.apply(x, lambd)`` -> a tensor equal to ``x`` whose gradient is ``-lambd * dL/dy``."""

    @staticmethod
    def forward(ctx, x: torch.Tensor, lambd: float) -> torch.Tensor:  # noqa: D102
        ctx.neg_scale = -float(lambd)
        return x.view(x.shape)

    @staticmethod
    def backward(ctx, grad_output: torch.Tensor) -> Tuple[Optional[torch.Tensor], None]:  # noqa: D102
        return (grad_output * ctx.neg_scale if ctx.needs_input_grad[0] else None), None


def grad_reverse(x: torch.Tensor, lambd: float = 1.0) -> torch.Tensor:
    """Functional form; ``lambd`` is the reversal strength (grad_reverse_dict of the model config)."""
    return GradReverse.apply(x, lambd)


class GradReverseLayer(nn.Module):
    """Module form, placed in front of an adversarial head (networks.py: CellDecoder)."""

    def __init__(self, lambd: float = 1.0) -> None:
        super().__init__()
        self.lambd = lambd

    def extra_repr(self) -> str:
        return f"lambd={self.lambd}"

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return GradReverse.apply(x, self.lambd)
