"""Optimizer + schedule of src/models/base_model.py:59-106 ("next" row 8f-1):
Adam wrapped by pl_bolts' LARSWrapper, LinearWarmupCosineAnnealingLR stepped per
optimizer step.  pl_bolts 0.2.2 is not vendored by the reference nor installed
here -- semantics restated from its published source, PARITY UNPINNED (checked
against the restatement in oracle/optim.py).  The whole parameter list is updated
by two HIP launches (``simhand_lars_adam_multi``: norm partials, then the fused
LARS+Adam update); ``multi_tensor=False`` keeps one launch group per tensor.
"""
from __future__ import annotations

import math
from typing import Iterable

import numpy as np
import torch
from torch.autograd.graph import increment_version

from .. import ops


class LARSAdam(torch.optim.Optimizer):
    """torch.optim.Adam(params, lr) optionally wrapped like LARSWrapper(eta=0.02, clip=True, eps=1e-8)."""

    def __init__(self, params: Iterable, lr: float, betas=(0.9, 0.999), eps: float = 1e-8, weight_decay: float = 0.0,
                 lars: bool = True, lars_eta: float = 0.02, lars_eps: float = 1e-8, lars_clip: bool = True,
                 multi_tensor: bool = True):
        defaults = dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, lars=lars, lars_eta=lars_eta,
                        lars_eps=lars_eps, lars_clip=lars_clip)
        super().__init__(params, defaults)
        self.multi_tensor = multi_tensor
        self._plans = {}
        # Loss-scaled (fp16) training: step(found_inf=<GradScaler's device flag>) lets the update launch skip itself on the device.  The
        # host-side step counters (Adam's bias corrections) must not advance over a skipped step: the flag is copied to pinned host
        # memory behind the launch and read at the NEXT step -- by then it has long arrived, the launch queue never drains.
        self._pending = None  # (pinned flag, event, parameters whose counters were advanced optimistically)

    accepts_found_inf = True

    def _resolve_pending(self, block: bool = True) -> None:
        if self._pending is None:
            return
        host, ev, ps = self._pending
        if not block and not ev.query():
            return  # the flag is still on its way: the next step() settles it
        self._pending = None
        ev.synchronize()
        if float(host[0]) != 0.0:  # that step was skipped on the device: take its count back
            for p in ps:
                self.state[p]["step"] -= 1

    # state[p]["step"] is PROVISIONAL between a guarded step and the next call of any method below (a step the device skipped is taken
    # back here, one step late); every entry point that exposes or replaces the counters settles it first
    def state_dict(self):
        self._resolve_pending()
        return super().state_dict()

    def load_state_dict(self, state_dict):
        self._resolve_pending()  # a flag still pending belongs to the OLD counters: settle it before they are replaced
        return super().load_state_dict(state_dict)

    def zero_grad(self, set_to_none: bool = True):
        # called at the top of every step, BEFORE the next forward / backward are enqueued: never block here (a blocking read would idle
        # the GPU once per iteration under precision 16); zero_grad neither exposes nor replaces the counters
        self._resolve_pending(block=False)
        return super().zero_grad(set_to_none=set_to_none)

    def _state(self, p):
        st = self.state[p]
        if not st:
            st["step"] = 0
            st["exp_avg"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
            st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
        return st

    @torch.no_grad()
    def step(self, closure=None, found_inf=None):
        self._resolve_pending()
        if found_inf is not None and not (self.multi_tensor and found_inf.is_cuda):
            if float(found_inf) != 0.0:  # per-tensor launches have no device-side guard: read the flag (one synchronisation)
                return
            found_inf = None
        if not self.multi_tensor:
            for group in self.param_groups:
                for p in group["params"]:
                    if p.grad is None:
                        continue
                    st = self._state(p)
                    st["step"] += 1
                    ops.lars_adam_step(p.data, p.grad.contiguous(), st["exp_avg"], st["exp_avg_sq"], st["step"], group["lr"],
                                       group["weight_decay"], group["lars"], group["betas"], group["eps"], group["lars_eta"],
                                       group["lars_eps"], group["lars_clip"])
                    increment_version(p)  # the kernel wrote through the raw pointer: packed-weight caches key on _version
            return
        # groups that share the kernel-wide constants go out together: the whole list in two launches
        buckets = {}
        for group in self.param_groups:
            key = (tuple(group["betas"]), group["eps"], group["lars_eta"], group["lars_eps"], bool(group["lars_clip"]))
            for p in group["params"]:
                if p.grad is not None:
                    buckets.setdefault(key, []).append((group, p))
        for key, items in buckets.items():
            betas, eps, eta, leps, clip = key
            pkey = (key, tuple(id(p) for _, p in items))
            plan = self._plans.get(pkey)
            if plan is None:
                plan = self._plans[pkey] = ops.LarsAdamPlan([p.numel() for _, p in items], items[0][1].device)
            rec = np.zeros(len(items), dtype=ops.OPT_TENSOR_DTYPE)
            keep = []  # contiguous gradient copies must outlive the launch
            for i, (group, p) in enumerate(items):
                st = self._state(p)
                st["step"] += 1
                g = p.grad if p.grad.is_contiguous() else p.grad.contiguous()
                keep.append(g)
                t = st["step"]
                rec[i] = (p.data_ptr(), g.data_ptr(), st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr(), p.numel(),
                          plan.first[i], plan.nchunks[i], group["lr"], group["weight_decay"],
                          np.float32(1.0) - np.float32(betas[0]) ** np.float32(t),
                          np.sqrt(np.float32(1.0) - np.float32(betas[1]) ** np.float32(t)), int(bool(group["lars"])), 0)
            ops.lars_adam_multi(plan, rec, betas, eps, eta, leps, clip, found_inf=found_inf)  # same stream as the producers of `keep`
            # the kernels wrote through raw pointers: tell torch, the packed-weight caches key on _version
            increment_version([p for _, p in items])
        if found_inf is not None and buckets:
            host = torch.empty(1, dtype=torch.float32, pin_memory=True)
            host.copy_(found_inf, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record()
            self._pending = (host, ev, [p for items in buckets.values() for _, p in items])


class LinearWarmupCosineAnnealingLR:
    """Closed form of pl_bolts' scheduler (warmup_start_lr -> base over
    ``warmup_epochs`` steps, then half-cosine to eta_min at ``max_epochs``)."""

    def __init__(self, optimizer, warmup_epochs: int, max_epochs: int, warmup_start_lr: float = 0.0, eta_min: float = 0.0):
        self.optimizer, self.warmup_epochs, self.max_epochs = optimizer, warmup_epochs, max_epochs
        self.warmup_start_lr, self.eta_min = warmup_start_lr, eta_min
        self.base_lrs = [g["lr"] for g in optimizer.param_groups]
        self.last_epoch = 0
        self._apply()

    def lr_at(self, t: int, base: float) -> float:
        if t < self.warmup_epochs:
            if self.warmup_epochs <= 1:
                return base
            return self.warmup_start_lr + t * (base - self.warmup_start_lr) / (self.warmup_epochs - 1)
        span = max(1, self.max_epochs - self.warmup_epochs)
        return self.eta_min + 0.5 * (base - self.eta_min) * (1 + math.cos(math.pi * (t - self.warmup_epochs) / span))

    def _apply(self):
        for g, b in zip(self.optimizer.param_groups, self.base_lrs):
            g["lr"] = self.lr_at(self.last_epoch, b)

    def step(self):
        self.last_epoch += 1
        self._apply()

    def get_last_lr(self):
        return [g["lr"] for g in self.optimizer.param_groups]

    def state_dict(self):
        return {"last_epoch": self.last_epoch, "base_lrs": self.base_lrs}

    def load_state_dict(self, s):
        self.last_epoch, self.base_lrs = s["last_epoch"], s["base_lrs"]
        self._apply()


class CosineAnnealingLR(LinearWarmupCosineAnnealingLR):
    """torch.optim.lr_scheduler.CosineAnnealingLR(T_max) closed form (optimizer='adam' branch, base_model.py:101-102)."""

    def __init__(self, optimizer, T_max: int):
        super().__init__(optimizer, 0, T_max, 0.0, 0.0)
