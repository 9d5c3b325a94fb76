"""``BaseModel`` with the reference's interface (src/models/base_model.py:12-127):
encoder construction, weight-decay param split (with the "bn"-substring quirk),
``setup`` (iters / epoch), ``configure_optimizers`` (Adam lr*sqrt(1024k) ->
LARS -> linear-warmup cosine, per step) and the epoch-end metric means."""
from __future__ import annotations

import math
from typing import Dict, Iterator, List, Tuple, Union

import torch

from .lightning import LightningModule
from .model_utils import get_wrapper_model
from .optim import CosineAnnealingLR, LARSAdam, LinearWarmupCosineAnnealingLR


class BaseModel(LightningModule):
    def __init__(self, config, logger_debug=None, mode: str = "train"):
        super().__init__()
        self.logger_debug = logger_debug
        self.mode = mode
        if "resnet_size" in config.keys():
            # the reference asks for ImageNet weights (pretrained=True, base_model.py:24); no network here
            self.encoder = get_wrapper_model(config, pretrained=True)
        self.config = config
        self.train_metrics_epoch: Dict[str, torch.Tensor] = {}
        self.train_metrics: Dict[str, torch.Tensor] = {}
        self.validation_metrics_epoch: Dict[str, torch.Tensor] = {}
        self.plot_params: dict = {}
        self.train_iters_per_epoch = 1

    def set_compute_dtype(self, dtype: torch.dtype, fp8: bool = False) -> None:
        """dtype: storage / kernel dtype of the encoder (fp32 parity mode or bf16); fp8: e4m3 forward operands for the
        matrix-core-bound layers on top of bf16 (BASELINE configs[4])."""
        self.encoder.set_compute_dtype(dtype, fp8=fp8)

    def exclude_from_wt_decay(self, named_params: Iterator[Tuple[str, torch.Tensor]], weight_decay: float,
                              skip_list: List[str] = ["bias", "bn"]) -> List[Dict[str, Union[list, float]]]:
        """base_model.py:32-53.  Substring match on the NAME: `features.1.*` (stem BN) and
        `downsample.1.*` BN weights do get weight decay -- kept."""
        params, excluded = [], []
        for name, p in named_params:
            if not p.requires_grad:
                continue
            (excluded if any(s in name for s in skip_list) else params).append(p)
        return [{"params": params, "weight_decay": weight_decay}, {"params": excluded, "weight_decay": 0.0}]

    def setup(self, stage: str = "fit"):
        world = self.trainer.world_size if self.trainer is not None else 1
        # base_model.py:55-57; under the reference's DP world_size is 1 and config.batch_size is the
        # global batch -- here config.batch_size stays the GLOBAL batch, so no extra factor.
        self.train_iters_per_epoch = max(1, self.config.num_samples // self.config.batch_size)
        self._world = world

    def configure_optimizers(self):
        groups = self.exclude_from_wt_decay(self.named_parameters(), weight_decay=self.config.opt_weight_decay)
        lr = self.config.lr * math.sqrt(1024 * self.config.num_of_mini_batch)
        k = self.config.num_of_mini_batch
        warmup = self.config.warmup_epochs * self.train_iters_per_epoch // k
        if "lr_max_epochs" in self.config.keys() and self.config["lr_max_epochs"] is not None:
            max_steps = self.config["lr_max_epochs"] * self.train_iters_per_epoch // k
        else:
            max_steps = self.trainer.max_epochs * self.train_iters_per_epoch // k
        lars = self.config.optimizer == "LARS"
        optimizer = LARSAdam(groups, lr=lr, lars=lars)
        if lars:
            scheduler = LinearWarmupCosineAnnealingLR(optimizer, warmup_epochs=warmup, max_epochs=max_steps,
                                                      warmup_start_lr=0, eta_min=0)
        else:
            scheduler = CosineAnnealingLR(optimizer, T_max=max_steps)
        return [optimizer], [{"scheduler": scheduler, "interval": "step", "frequency": 1}]

    def training_epoch_end(self, outputs: List[dict]):
        keys = outputs[0].keys()
        self.train_metrics_epoch = {k: torch.stack([x[k] for x in outputs]).mean() for k in keys}
        self.log("checkpoint_saving_loss", self.train_metrics_epoch["loss_3d" if "loss_3d" in keys else "loss"])

    def validation_epoch_end(self, outputs: List[dict]):
        keys = outputs[0].keys()
        self.validation_metrics_epoch = {k: torch.stack([x[k] for x in outputs]).mean() for k in keys}
