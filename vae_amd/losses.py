"""Alternative classification losses and GroupDRO re-weighting with the reference's names and
call signatures (src/losses.py:9-202).  Off by default in the reference (task.py:133-134,
config.py:80) and tiny (B x n_classes): plain torch, no kernels."""
from __future__ import annotations

from typing import List, Optional

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F


def _smoothed_targets(target: torch.Tensor, num_classes: int) -> torch.Tensor:
    """Label smoothing used by both focal losses: binary 0.95/0.05; otherwise 0.9 on the class
    and 0.05 on each neighbour (clipped at the ends, so an end class keeps 0.95)."""
    target = target.clamp(0, num_classes - 1)
    if num_classes == 2:
        hot = F.one_hot(target, num_classes)
        return 0.95 * hot + 0.05 * (1 - hot)
    out = 0.9 * F.one_hot(target, num_classes)
    for step in (-1, 1):
        out = out + 0.05 * F.one_hot((target + step).clamp(0, num_classes - 1), num_classes)
    return out


class Poly1FocalLoss(nn.Module):
    def __init__(self, num_classes: int, epsilon: float = 1.0, alpha: float = 0.25, gamma: float = 2.0,
                 reduction: str = "none", weight: torch.Tensor = None, pos_weight: torch.Tensor = None,
                 label_is_onehot: bool = False):
        super().__init__()
        self.num_classes, self.epsilon, self.alpha, self.gamma = num_classes, epsilon, alpha, gamma
        self.reduction, self.weight, self.pos_weight, self.label_is_onehot = reduction, weight, pos_weight, label_is_onehot

    def _convert_target(self, target):
        return _smoothed_targets(target, self.num_classes)

    def forward(self, logits, labels):
        if not self.label_is_onehot:
            if labels.ndim == 1:
                labels = self._convert_target(labels)
            else:
                labels = F.one_hot(labels.unsqueeze(1), self.num_classes).transpose(1, -1).squeeze_(-1)
        labels = labels.to(device=logits.device, dtype=logits.dtype)
        p = torch.sigmoid(logits)
        bce = F.binary_cross_entropy_with_logits(logits, labels, reduction="none", weight=self.weight,
                                                 pos_weight=self.pos_weight)
        one_minus_pt = 1 - (labels * p + (1 - labels) * (1 - p))
        focal = bce * one_minus_pt ** self.gamma
        if self.alpha >= 0:
            focal = (self.alpha * labels + (1 - self.alpha) * (1 - labels)) * focal
        poly1 = (focal + self.epsilon * one_minus_pt ** (self.gamma + 1)).sum(dim=-1)
        if self.reduction == "mean":
            return poly1.mean()
        if self.reduction == "sum":
            return poly1.sum()
        return poly1


class FocalLoss(nn.Module):
    def __init__(self, num_classes: int, gamma: float = 2.0, alpha: float = 1.0):
        super().__init__()
        self.gamma, self.num_classes, self.alpha = gamma, num_classes, alpha

    def _convert_target(self, target):
        return _smoothed_targets(target, self.num_classes)

    def forward(self, input, target):
        logp = F.log_softmax(input, dim=-1)
        loss = -self._convert_target(target) * logp * (1 - logp.exp()) ** self.gamma
        return self.alpha * loss.sum(dim=-1)


class GroupDRO:
    """Running per-group loss -> per-sample weights, balanced within each list of groups."""

    DEFAULT_GROUPS = [0, 1, 10, 11, 20, 21, 30, 31, 40, 41, 50, 51, 60, 61, 998, 999]
    DEFAULT_BALANCES = [[0, 1, 60, 61], [10, 11, 50, 51], [20, 21, 40, 41], [30, 31]]

    def __init__(self, groups: Optional[List] = None, balances: Optional[List] = None, alpha: float = 0.99):
        self.groups = list(self.DEFAULT_GROUPS) if groups is None else groups
        self.alpha = alpha
        self.batch_loss = dict.fromkeys(self.groups, 0.0)
        self.batch_count = dict.fromkeys(self.groups, 0)
        self.running_loss = dict.fromkeys(self.groups, 0.0)
        self.weights = dict.fromkeys(self.groups, 1.0)
        self.balances = [list(b) for b in self.DEFAULT_BALANCES] if balances is None else balances

    def update_batch(self, loss, mask, group_idx):
        for l, m, g in zip(loss.tolist(), mask.tolist(), group_idx):
            self.batch_loss[g] += l * m
            self.batch_count[g] += m

    def update_weights(self):
        for g in self.groups:
            if self.batch_count[g] > 1:
                batch_mean = self.batch_loss[g] / self.batch_count[g]
                self.running_loss[g] = self.alpha * self.running_loss[g] + (1 - self.alpha) * batch_mean
            self.batch_loss[g], self.batch_count[g] = 0.0, 0
        for members in self.balances:
            mean_w = np.mean(np.array([self.running_loss[g] for g in members]))
            for g in members:
                self.weights[g] = self.running_loss[g] / mean_w

    def get_weights(self, group_idx):
        return torch.from_numpy(np.array([self.weights[g] for g in group_idx]))
