"""Dynamic loss scaling for the fp16-storage mode -- the semantics of ``torch.cuda.amp.GradScaler`` that the reference trains under
(``Trainer(precision=16, amp_backend="native")``, src/experiments/main.py:158-159; PyTorch 1.12 defaults: init_scale 2^16, growth 2x
every 2000 clean steps, backoff 0.5x and a skipped optimizer step on inf / nan).

fp16 keeps 11 significand bits (the forward activations are what decides how closely a 16-bit run follows fp32:
profiles/r03_stability_160steps.md) but only 5 exponent bits, so the backward pass runs on ``scale * loss``: every gradient tensor the
HIP kernels store in fp16 is ``scale`` times larger, the fp32 parameter gradients are unscaled before the optimizer sees them, and a
step whose gradients overflowed is skipped.  The non-finite check / unscale is torch's own fused foreach op on the device
(plumbing); reading its flag is the one host synchronisation per step, as in torch's GradScaler.step."""
from __future__ import annotations

from typing import Iterable

import torch


class GradScaler:
    def __init__(self, init_scale: float = 2.0 ** 16, growth_factor: float = 2.0, backoff_factor: float = 0.5, growth_interval: int = 2000,
                 enabled: bool = True):
        self.enabled = enabled
        self._scale, self.growth_factor, self.backoff_factor, self.growth_interval = float(init_scale), growth_factor, backoff_factor, growth_interval
        self._growth_tracker = 0
        self._found_inf = None
        self.skipped_steps = 0

    def get_scale(self) -> float:
        return self._scale if self.enabled else 1.0

    def scale(self, loss: torch.Tensor) -> torch.Tensor:
        return loss * self._scale if self.enabled else loss

    def unscale_(self, params: Iterable[torch.nn.Parameter]) -> None:
        """grad <- grad / scale for every parameter gradient, recording whether any of them holds an inf / nan."""
        if not self.enabled:
            return
        grads = [p.grad for p in params if p.grad is not None]
        if not grads:
            return
        dev = grads[0].device
        found = torch.zeros(1, dtype=torch.float32, device=dev)
        inv = torch.full((1,), 1.0 / self._scale, dtype=torch.float32, device=dev)
        torch._amp_foreach_non_finite_check_and_unscale_(grads, found, inv)
        self._found_inf = found

    def step(self, optimizer, *args, **kwargs) -> bool:
        """optimizer.step() unless the unscaled gradients were non-finite; returns whether the step ran."""
        if self.enabled and self._found_inf is not None and float(self._found_inf) != 0.0:
            self.skipped_steps += 1
            return False
        optimizer.step(*args, **kwargs)
        return True

    def update(self) -> None:
        if not self.enabled:
            return
        if self._found_inf is not None and float(self._found_inf) != 0.0:
            self._scale *= self.backoff_factor
            self._growth_tracker = 0
        else:
            self._growth_tracker += 1
            if self._growth_tracker == self.growth_interval:
                self._scale *= self.growth_factor
                self._growth_tracker = 0
        self._found_inf = None

    def state_dict(self) -> dict:
        return {"scale": self._scale, "growth_factor": self.growth_factor, "backoff_factor": self.backoff_factor,
                "growth_interval": self.growth_interval, "_growth_tracker": self._growth_tracker}

    def load_state_dict(self, state: dict) -> None:
        self._scale = float(state["scale"])
        self.growth_factor, self.backoff_factor = state["growth_factor"], state["backoff_factor"]
        self.growth_interval, self._growth_tracker = state["growth_interval"], state["_growth_tracker"]
