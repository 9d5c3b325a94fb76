"""Minimal stand-ins for the pytorch_lightning 1.8 pieces the reference's hot
path touches (pytorch_lightning is not installed in this image and is not part
of the hot path): ``LightningModule`` (nn.Module + ``log`` /
``save_hyperparameters`` / ``trainer``), ``seed_everything``,
``ModelCheckpoint`` (monitor / mode / save_top_k / filename template of
src/experiments/main.py:143-149) and a ``Trainer`` whose ``fit`` runs the
training_step -> backward -> optimizer/scheduler step loop with one process
per GPU (RCCL) instead of the reference's ``strategy="dp"``.
"""
from __future__ import annotations

import collections
import os
import random
import re
from typing import Any, Deque, Dict, List, Optional

import numpy as np
import torch
import torch.distributed as dist
from torch import nn


def seed_everything(seed: int) -> int:
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)
    os.environ["PL_GLOBAL_SEED"] = str(seed)
    return seed


class LightningModule(nn.Module):
    def __init__(self, *a, **k):
        super().__init__()
        self.trainer: Optional["Trainer"] = None
        self.logged: Dict[str, Any] = {}
        self.hparams: Dict[str, Any] = {}

    def log(self, name: str, value, *a, **k) -> None:
        self.logged[name] = value

    def save_hyperparameters(self, *a, **k) -> None:
        cfg = getattr(self, "config", None)
        if cfg is not None:
            self.hparams = {"config": dict(cfg)}


class ModelCheckpoint:
    """monitor='contrastive_loss', mode='min', save_top_k, filename template
    with {epoch:02d} and {contrastive_loss:.6f} (main.py:143-149).  Files are
    Lightning-style dicts: state_dict, hyper_parameters, optimizer_states,
    lr_schedulers, epoch, global_step."""

    def __init__(self, save_top_k: int = 3, monitor: str = "contrastive_loss", mode: str = "min",
                 filename: Optional[str] = None, dirpath: Optional[str] = None):
        self.save_top_k, self.monitor, self.mode = save_top_k, monitor, mode
        self.filename = filename or "{epoch:02d}"
        self.dirpath = dirpath
        self.best_model_path = ""
        self.kept: List = []  # (score, path)

    def format_name(self, epoch: int, metrics: Dict[str, float]) -> str:
        name = self.filename
        name = name.replace("{epoch:02d}", f"epoch={epoch:02d}")
        for k, v in metrics.items():
            name = name.replace("{" + k + ":.6f}", f"{k}={v:.6f}")
        return name + ".ckpt"

    def on_epoch_end(self, trainer: "Trainer", module: LightningModule, epoch: int, metrics: Dict[str, float]) -> None:
        if self.save_top_k == 0 or self.monitor not in metrics or trainer.global_rank != 0:
            return
        score = metrics[self.monitor]
        key = score if self.mode == "min" else -score
        if self.save_top_k > 0 and len(self.kept) >= self.save_top_k and key >= max(s for s, _ in self.kept):
            return
        os.makedirs(self.dirpath, exist_ok=True)
        path = os.path.join(self.dirpath, self.format_name(epoch, metrics))
        # a resumed run may rewrite a file it already lists (default template '{epoch:02d}', or a mid-epoch resume that re-runs the
        # epoch): ONE entry per path, the new score replaces the old one
        self.kept = [(k, p) for k, p in self.kept if p != path]
        self.kept.append((key, path))
        self.kept.sort(key=lambda t: t[0])
        drops = []
        while self.save_top_k > 0 and len(self.kept) > self.save_top_k:
            drops.append(self.kept.pop()[1])
        self.best_model_path = self.kept[0][1]
        torch.save(trainer.checkpoint_dict(module, epoch), path)  # records `kept` INCLUDING itself: what a resumed run may prune later
        live = {p for _, p in self.kept}
        for drop in drops:
            if drop not in live and os.path.exists(drop):  # never remove a file an entry (or best_model_path) still points to
                os.remove(drop)

    def _template_regex(self) -> "re.Pattern":
        """This callback's filename template as a regular expression (literal text kept, every {name:fmt} field a number)."""
        parts = re.split(r"(\{[^{}]+\})", self.filename)
        rx = "".join((re.escape(f[1:-1].split(":")[0]) + r"=(?P<%s>-?\d+(?:\.\d+)?)" % re.sub(r"\W", "_", f[1:-1].split(":")[0]))
                     if f.startswith("{") else re.escape(f) for f in parts)
        return re.compile("^" + rx + r"\.ckpt$")

    def restore(self, kept, best: str = "") -> None:
        """Adopt the (score key, path) list a checkpoint of THIS run recorded: exact scores, and only files the run wrote itself -- and
        only those in THIS callback's directory (Lightning does the same: a resumed run with another dirpath starts its own top-k and
        never prunes the run it was started from)."""
        here = os.path.abspath(self.dirpath) if self.dirpath else None
        self.kept = sorted(((float(k), str(p)) for k, p in kept
                            if os.path.exists(p) and (here is None or os.path.abspath(os.path.dirname(p)) == here)), key=lambda t: t[0])
        self.best_model_path = self.kept[0][1] if self.kept else ""

    def rescan(self) -> None:
        """Fallback for checkpoints that carry no kept-list (written before round 4): rebuild `kept` from the files in dirpath whose
        WHOLE name matches this callback's filename template -- other runs' files that merely contain "{monitor}=<float>" are never
        adopted, hence never pruned.  Scores come back at the template's 6 decimals."""
        if not self.dirpath or not os.path.isdir(self.dirpath):
            return
        rx = self._template_regex()
        field = re.sub(r"\W", "_", self.monitor)
        known = {p for _, p in self.kept}
        for fn in sorted(os.listdir(self.dirpath)):
            m = rx.match(fn)
            path = os.path.join(self.dirpath, fn)
            if m is None or field not in m.groupdict() or path in known:
                continue
            score = float(m.group(field))
            self.kept.append((score if self.mode == "min" else -score, path))
        self.kept.sort(key=lambda t: t[0])
        if self.kept:
            self.best_model_path = self.kept[0][1]


class Trainer:
    """fit loop for the contrastive step classes.  Mixed-precision policy (the reference runs fp16 autocast +
    GradScaler, main.py:158-159):
      precision 32        -> exact-fp32 MFMA kernels, fp32 storage (parity mode);
      precision 16        -> THE REFERENCE'S POLICY: fp16 storage (fp16 build of the library: 11-bit significand) + fp16 MFMA, fp32
                             accumulators / BatchNorm statistics / loss / optimizer / master weights, dynamic loss scaling with
                             GradScaler semantics (host/amp.py: 2^16, x0.5 and a skipped step on overflow, x2 every 2000 clean steps);
      precision "bf16"    -> bf16 storage + bf16 MFMA, otherwise the same (BASELINE's benchmark dtype).  No loss scaling: bf16 has
                             fp32's exponent range.  Its 8-bit significand is what profiles/r03_stability_160steps.md shows as a
                             ~10 % slower-learning run on the noise-memorisation task, all of it from the FORWARD activations'
                             rounding -- which is why 16 maps to fp16, as in the reference, and not to bf16;
      precision "fp8"     -> as bf16, plus e4m3 forward operands (per-tensor scales: weights current, activations delayed
                             with a 16-entry amax ring and 1 bit of margin, ops.FP8Scaler) for the matrix-core-bound layers.
    tests/test_gpu_fp8.py / test_gpu_main.py and profiles/r02_stability_160steps.md (scripts/stability_run.py: 160 steps, fp32 mode ==
    the oracle's curve, bf16 mode == the oracle's bf16-storage twin) hold the multi-step stability evidence.
    Resume: bf16 / fp32 runs continue bit for bit (tests/test_gpu_main.py); the GradScaler state (precision 16) and the fp8
    delayed-scaling amax rings travel in the checkpoint ("native_amp_scaling_state", "fp8_scaling_state")."""

    def __init__(self, max_epochs: int = 1, precision=32, callbacks: Optional[list] = None, log_every_n_steps: int = 5,
                 default_root_dir: str = ".", max_steps: int = -1, logger=None, keep_step_losses: int = 4096, sync_batchnorm: bool = False,
                 **_ignored):
        self.max_epochs, self.precision, self.callbacks = max_epochs, precision, callbacks or []
        self.sync_batchnorm = bool(sync_batchnorm)  # PL's Trainer(sync_batchnorm=...): BatchNorm statistics over the global batch
        self.log_every_n_steps, self.default_root_dir, self.max_steps = log_every_n_steps, default_root_dir, max_steps
        self.global_step = 0
        self.current_epoch = 0
        self._epoch_complete, self._batches_seen = True, 0
        # detached per-step loss scalars (device tensors: no host sync per step); bounded -- a full run is 1e5..1e6 steps and every
        # entry pins an allocator block
        self.step_losses: Deque[torch.Tensor] = collections.deque(maxlen=max(1, keep_step_losses))
        self.history: List[Dict[str, float]] = []
        self.optimizers: list = []
        self.schedulers: list = []
        from .amp import GradScaler

        self.scaler = GradScaler(enabled=str(precision) == "16")  # the reference's native-AMP loss scaling (fp16 storage only)

    @property
    def world_size(self) -> int:
        return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1

    @property
    def global_rank(self) -> int:
        return dist.get_rank() if dist.is_available() and dist.is_initialized() else 0

    @property
    def compute_dtype(self) -> torch.dtype:
        p = str(self.precision)
        if p == "32":
            return torch.float32
        return torch.float16 if p == "16" else torch.bfloat16  # "bf16", "fp8"

    @property
    def fp8(self) -> bool:
        return str(self.precision) == "fp8"

    def checkpoint_dict(self, module: LightningModule, epoch: int) -> dict:
        return {
            "epoch": epoch, "global_step": self.global_step, "pytorch-lightning_version": "1.8.0-compatible",
            # where inside `epoch` the run stood: a checkpoint taken when max_steps ended an epoch early resumes
            # in the SAME epoch after skipping the batches already consumed
            "epoch_complete": self._epoch_complete, "batches_seen": self._batches_seen,
            "state_dict": module.state_dict(), "hyper_parameters": module.hparams,
            "optimizer_states": [o.state_dict() for o in self.optimizers],
            "lr_schedulers": [s["scheduler"].state_dict() for s in self.schedulers],
            # the GradScaler state under BOTH of Lightning's keys: 1.8 (the reference's pin) writes the precision plugin's state under
            # its class name; "native_amp_scaling_state" is the pre-1.6 key 1.8 still reads for back-compat
            "NativeMixedPrecisionPlugin": self.scaler.state_dict() if self.scaler.enabled else None,
            "native_amp_scaling_state": self.scaler.state_dict() if self.scaler.enabled else None,
            # the files this run's ModelCheckpoint callbacks wrote and still keep (a resumed run prunes these, and only these)
            "callbacks_kept": [[list(map(list, cb.kept)), cb.best_model_path] for cb in self.callbacks if isinstance(cb, ModelCheckpoint)],
            # fp8 configuration: the delayed-scaling amax rings of every quantisation site (a resumed run continues with the same scales)
            "fp8_scaling_state": (module.encoder.engine.fp8_state_dict() if self.fp8 and hasattr(getattr(module, "encoder", None), "engine") else None),
        }

    def fit(self, model: LightningModule, train_dataloaders=None, val_dataloaders=None, ckpt_path: Optional[str] = None):
        from .dist import OverlappedGradReducer, allreduce_gradients, broadcast_module_state

        model.trainer = self
        if hasattr(model, "set_compute_dtype"):
            model.set_compute_dtype(self.compute_dtype, fp8=self.fp8)
        device = torch.device("cuda", torch.cuda.current_device())
        model.to(device)
        model.setup("fit")
        opts, scheds = model.configure_optimizers()
        self.optimizers, self.schedulers = opts, scheds
        start_epoch, skip_batches = 0, 0
        resume_kept = []
        if ckpt_path:
            ck = torch.load(ckpt_path, map_location=device, weights_only=False)
            resume_kept = list(ck.get("callbacks_kept") or [])
            model.load_state_dict(ck["state_dict"])
            for o, s in zip(opts, ck.get("optimizer_states", [])):
                o.load_state_dict(s)
            for s, st in zip(scheds, ck.get("lr_schedulers", [])):
                s["scheduler"].load_state_dict(st)
            if self.scaler.enabled:
                amp_state = ck.get("NativeMixedPrecisionPlugin") or ck.get("native_amp_scaling_state")  # PL 1.8's key first
                if amp_state:
                    self.scaler.load_state_dict(amp_state)
                else:
                    import warnings

                    warnings.warn(f"{ckpt_path}: precision 16 resume without a GradScaler state (neither 'NativeMixedPrecisionPlugin' nor "
                                  "'native_amp_scaling_state'): loss scaling restarts at its initial scale")
            if self.fp8 and ck.get("fp8_scaling_state") and hasattr(getattr(model, "encoder", None), "engine"):
                model.encoder.engine.load_fp8_state_dict(ck["fp8_scaling_state"], device)
            self.global_step = ck.get("global_step", 0)
            if ck.get("epoch_complete", True):
                start_epoch = ck.get("epoch", -1) + 1
            else:
                start_epoch, skip_batches = ck.get("epoch", 0), ck.get("batches_seen", 0)
        # replicas start from rank 0's parameters / buffers whatever each rank's seeding or checkpoint file was
        broadcast_module_state(model)
        for cb in self.callbacks:
            if isinstance(cb, ModelCheckpoint) and cb.dirpath is None:
                cb.dirpath = os.path.join(self.default_root_dir, "checkpoints")
            if isinstance(cb, ModelCheckpoint) and ckpt_path:
                # earlier top-k files of the run being resumed stay subject to pruning: the list the checkpoint itself carries (exact
                # scores, this run's files only); checkpoints without one fall back to a scan for this callback's filename template
                rec = resume_kept.pop(0) if resume_kept else None
                if rec is not None:
                    cb.restore(rec[0], rec[1])
                else:
                    cb.rescan()
        reducer = None
        if self.sync_batchnorm:
            from .dist import enable_sync_bn

            enable_sync_bn(getattr(model, "process_group", None))
        if self.world_size > 1 and hasattr(getattr(model, "encoder", None), "engine"):
            reducer = OverlappedGradReducer(getattr(model, "process_group", None))
            model.encoder.engine.grad_reducer = reducer  # backbone gradients are all-reduced during the backward pass
        model.train()
        for epoch in range(start_epoch, self.max_epochs):
            self.current_epoch = epoch
            outputs = []
            self._epoch_complete, self._batches_seen = False, 0
            for batch_idx, batch in enumerate(train_dataloaders):
                self._batches_seen = batch_idx + 1
                if epoch == start_epoch and batch_idx < skip_batches:
                    continue  # consumed before the checkpoint was taken (the iterator still advanced its generator)
                batch = {k: (v.to(device, non_blocking=True) if torch.is_tensor(v) else v) for k, v in batch.items()}
                out = model.training_step(batch, batch_idx)
                loss = out["loss"]
                for o in opts:
                    o.zero_grad(set_to_none=True)
                self.scaler.scale(loss).backward()
                if self.world_size > 1:
                    allreduce_gradients(model.parameters(), group=getattr(model, "process_group", None),
                                        skip=reducer.reduced if reducer is not None else None)
                self.scaler.unscale_(model.parameters())
                for o in opts:
                    self.scaler.step(o)  # skipped when the fp16 gradients overflowed (every rank sees the same reduced gradients)
                self.scaler.update()
                for s in scheds:
                    if s.get("interval", "epoch") == "step":
                        s["scheduler"].step()
                self.global_step += 1
                self.step_losses.append(loss.detach())
                outputs.append({k: v.detach() for k, v in out.items() if torch.is_tensor(v)})
                if self.global_rank == 0 and self.global_step % self.log_every_n_steps == 0:
                    print(f"epoch {epoch} step {self.global_step} contrastive_loss {float(loss):.6f}", flush=True)
                if 0 < self.max_steps <= self.global_step:
                    break
            else:
                self._epoch_complete = True
            if not outputs:  # resumed exactly at an epoch's end
                continue
            model.training_epoch_end(outputs)
            metrics = {k: float(v) for k, v in model.train_metrics_epoch.items()}
            metrics["contrastive_loss"] = metrics.get("loss", float("nan"))
            self.history.append(metrics)
            for cb in self.callbacks:
                if hasattr(cb, "on_epoch_end"):
                    cb.on_epoch_end(self, model, epoch, metrics)
            for s in scheds:
                if s.get("interval", "epoch") == "epoch":
                    s["scheduler"].step()
            if 0 < self.max_steps <= self.global_step:
                break
        return self
