"""Restatement of the optimizer pieces the reference pulls from pl_bolts 0.2.2
(call site src/models/base_model.py:59-106).  TEST INFRASTRUCTURE ONLY.

PARITY UNPINNED: pl_bolts is not vendored in /root/reference and is not
installed here, so nothing can be executed against it; the classes below
restate the algorithm of its published source
(pl_bolts/optimizers/lars_scheduling.py, pl_bolts/optimizers/lr_scheduler.py,
version 0.2.2 as pinned by environment.yml:143).
"""
from __future__ import annotations

import math
from typing import List

import torch


class LARSWrapperOracle:
    """LARS trust-ratio scaling applied to the gradients in place, then the
    wrapped optimizer's own step with its weight decay switched off."""

    def __init__(self, optimizer: torch.optim.Optimizer, eta: float = 0.02, clip: bool = True, eps: float = 1e-8):
        self.optim = optimizer
        self.eta, self.clip, self.eps = eta, clip, eps
        self.param_groups = optimizer.param_groups

    @torch.no_grad()
    def step(self) -> None:
        saved = []
        for group in self.optim.param_groups:
            wd = group.get("weight_decay", 0)
            saved.append(wd)
            group["weight_decay"] = 0
            for p in group["params"]:
                if p.grad is None:
                    continue
                p_norm = torch.norm(p.data)
                g_norm = torch.norm(p.grad.data)
                if p_norm != 0 and g_norm != 0:
                    new_lr = (self.eta * p_norm) / (g_norm + p_norm * wd + self.eps)
                    if self.clip:
                        new_lr = min(new_lr / group["lr"], 1)
                    p.grad.data += wd * p.data
                    p.grad.data *= new_lr
        self.optim.step()
        for group, wd in zip(self.optim.param_groups, saved):
            group["weight_decay"] = wd


def linear_warmup_cosine_lr(step: int, base_lr: float, warmup_steps: int, max_steps: int,
                            warmup_start_lr: float = 0.0, eta_min: float = 0.0) -> float:
    """Closed form of LinearWarmupCosineAnnealingLR at scheduler step ``step``
    (0-based; the reference steps it once per optimizer step,
    base_model.py:104)."""
    if step < warmup_steps:
        if warmup_steps <= 1:
            return base_lr
        return warmup_start_lr + step * (base_lr - warmup_start_lr) / (warmup_steps - 1)
    return eta_min + 0.5 * (base_lr - eta_min) * (1 + math.cos(math.pi * (step - warmup_steps) / (max_steps - warmup_steps)))


def exclude_from_wt_decay(names: List[str], skip=("bias", "bn")):
    """src/models/base_model.py:32-53: substring test on the parameter NAME
    (misses features.1.* and downsample.1.* BN weights -- quirk kept)."""
    decay, no_decay = [], []
    for n in names:
        (no_decay if any(s in n for s in skip) else decay).append(n)
    return decay, no_decay
