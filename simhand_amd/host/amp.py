"""Dynamic loss scaling for the fp16-storage mode -- the semantics of ``torch.cuda.amp.GradScaler`` that the reference trains under
(``Trainer(precision=16, amp_backend="native")``, src/experiments/main.py:158-159; PyTorch 1.12 defaults: init_scale 2^16, growth 2x
every 2000 clean steps, backoff 0.5x and a skipped optimizer step on inf / nan).

fp16 keeps 11 significand bits (the forward activations are what decides how closely a 16-bit run follows fp32:
profiles/r03_stability_160steps.md) but only 5 exponent bits, so the backward pass runs on ``scale * loss``: every gradient tensor the
HIP kernels store in fp16 is ``scale`` times larger, the fp32 parameter gradients are unscaled before the optimizer sees them, and a
step whose gradients overflowed is skipped.

No host synchronisation per step: the scale, the growth tracker and the overflow flag live ON THE DEVICE (as in torch's GradScaler
with a fused optimizer).  The non-finite check / unscale and the scale update are torch's own fused device ops (plumbing); an optimizer
that takes ``found_inf`` (host/optim.py LARSAdam: the update kernel returns without touching anything when the flag is set) is handed the
device flag, any other optimizer falls back to reading it (one synchronisation, torch's own behaviour for unfused optimizers).  Host
reads (``get_scale``, ``skipped_steps``, ``state_dict``) synchronise when they are asked for, not per step."""
from __future__ import annotations

from typing import Iterable, Optional

import torch


class GradScaler:
    def __init__(self, init_scale: float = 2.0 ** 16, growth_factor: float = 2.0, backoff_factor: float = 0.5, growth_interval: int = 2000,
                 enabled: bool = True):
        self.enabled = enabled
        self.growth_factor, self.backoff_factor, self.growth_interval = growth_factor, backoff_factor, growth_interval
        self._init = (float(init_scale), 0, 0)  # (scale, growth tracker, skipped steps) until the device tensors exist
        self._scale_t: Optional[torch.Tensor] = None   # fp32 [1]
        self._growth_t: Optional[torch.Tensor] = None  # int32 [1]
        self._skipped_t: Optional[torch.Tensor] = None  # fp32 [1]: optimizer steps skipped so far
        self._found_inf: Optional[torch.Tensor] = None

    # -- device state ---------------------------------------------------------------------------------------------------------
    def _lazy(self, device) -> None:
        if self._scale_t is None or self._scale_t.device != device:
            sc, gt, sk = (self._scale, self._growth_tracker, self.skipped_steps) if self._scale_t is not None else self._init
            self._scale_t = torch.full((1,), sc, dtype=torch.float32, device=device)
            self._growth_t = torch.full((1,), gt, dtype=torch.int32, device=device)
            self._skipped_t = torch.full((1,), float(sk), dtype=torch.float32, device=device)

    # host views of the device state (each read synchronises; assignments rewrite the device value)
    @property
    def _scale(self) -> float:
        return float(self._scale_t) if self._scale_t is not None else self._init[0]

    @_scale.setter
    def _scale(self, v: float) -> None:
        if self._scale_t is not None:
            self._scale_t.fill_(float(v))
        else:
            self._init = (float(v), self._init[1], self._init[2])

    @property
    def _growth_tracker(self) -> int:
        return int(self._growth_t) if self._growth_t is not None else self._init[1]

    @_growth_tracker.setter
    def _growth_tracker(self, v: int) -> None:
        if self._growth_t is not None:
            self._growth_t.fill_(int(v))
        else:
            self._init = (self._init[0], int(v), self._init[2])

    @property
    def skipped_steps(self) -> int:
        return int(round(float(self._skipped_t))) if self._skipped_t is not None else self._init[2]

    def get_scale(self) -> float:
        return self._scale if self.enabled else 1.0

    # -- the four calls of a training step ----------------------------------------------------------------------------------------
    def scale(self, loss: torch.Tensor) -> torch.Tensor:
        if not self.enabled:
            return loss
        self._lazy(loss.device)
        return loss * self._scale_t[0]

    def unscale_(self, params: Iterable[torch.nn.Parameter]) -> None:
        """grad <- grad / scale for every parameter gradient, recording (on the device) whether any of them holds an inf / nan."""
        if not self.enabled:
            return
        grads = [p.grad for p in params if p.grad is not None]
        if not grads:
            return
        dev = grads[0].device
        self._lazy(dev)
        found = torch.zeros(1, dtype=torch.float32, device=dev)
        inv = self._scale_t.double().reciprocal().float()
        torch._amp_foreach_non_finite_check_and_unscale_(grads, found, inv)
        self._found_inf = found

    def step(self, optimizer, *args, **kwargs):
        """optimizer.step() unless the unscaled gradients were non-finite.  An optimizer that takes the device flag
        (``accepts_found_inf``) decides on the device -- no synchronisation, nothing returned; otherwise the flag is read here and
        the return value says whether the step ran."""
        if not self.enabled or self._found_inf is None:
            optimizer.step(*args, **kwargs)
            return True
        if getattr(optimizer, "accepts_found_inf", False):
            optimizer.step(*args, found_inf=self._found_inf, **kwargs)
            return None
        if float(self._found_inf) != 0.0:
            return False
        optimizer.step(*args, **kwargs)
        return True

    def update(self) -> None:
        if not self.enabled or self._found_inf is None:
            return
        self._skipped_t += (self._found_inf != 0).to(torch.float32)
        # found: scale *= backoff, tracker = 0;  else tracker += 1 and, at growth_interval, scale *= growth, tracker = 0
        torch._amp_update_scale_(self._scale_t, self._growth_t, self._found_inf, self.growth_factor, self.backoff_factor, self.growth_interval)
        self._found_inf = None

    # -- checkpoints ----------------------------------------------------------------------------------------------------------------
    def state_dict(self) -> dict:
        return {"scale": self._scale, "growth_factor": self.growth_factor, "backoff_factor": self.backoff_factor,
                "growth_interval": self.growth_interval, "_growth_tracker": self._growth_tracker}

    def load_state_dict(self, state: dict) -> None:
        self.growth_factor, self.backoff_factor = state["growth_factor"], state["backoff_factor"]
        self.growth_interval = state["growth_interval"]
        self._scale = float(state["scale"])
        self._growth_tracker = int(state["_growth_tracker"])
