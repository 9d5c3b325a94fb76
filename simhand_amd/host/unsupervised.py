"""Step classes with the reference's names and interface
(src/models/unsupervised/{simclr,simclr_w,peclr,peclr_w,simhand,simhand_base,
simhand_w,simhand_vis}_model.py): constructor ``Cls(config, logger_debug, mode)``,
``training_step`` / ``validation_step`` / ``contrastive_step`` /
``get_transformed_projections`` / ``get_adaptive_weights`` /
``get_projection_stats`` / ``get_encodings`` / ``forward``, attributes
``encoder``, ``projection_head`` (Sequential idx 0,1,2,3), ``config``,
``train_metrics``, ``plot_params``; ``self.log("contrastive_loss", ...)``.

The modules inside ``projection_head`` are parameter containers; the head runs
through the C ABI in fp32 (Linear = 1x1 implicit GEMM with fused BatchNorm1d
partial sums -> BN finalize -> BN-apply+ReLU -> Linear), then one fused
post-process kernel and the sharded fused NT-Xent.
"""
from __future__ import annotations

from typing import Dict, Optional, Tuple

import torch
from torch import Tensor, nn

from .. import ops
from .base_model import BaseModel
from .dist_loss import LossConfig
from . import model_utils as mu


class _HeadFn(torch.autograd.Function):
    """Linear(C,H,bias) -> BatchNorm1d(H) -> ReLU -> Linear(H,O,no bias), fp32."""

    @staticmethod
    def forward(ctx, enc, head: nn.Sequential, training: bool, w1, b1, gamma, beta, w2):
        lin1, bn, _, lin2 = head[0], head[1], head[2], head[3]
        n, c = enc.shape
        hdim, odim = w1.shape[0], w2.shape[0]
        f32 = torch.float32
        enc = enc.contiguous()
        d1 = ops.conv_desc(n, 1, 1, c, hdim, 1, 1, 1, 0, f32)
        d2 = ops.conv_desc(n, 1, 1, hdim, odim, 1, 1, 1, 0, f32)
        w1c, w2c = w1.detach().contiguous(), w2.detach().contiguous()  # [out][in] == KRSC for 1x1
        h, part = ops.conv2d_fwd(d1, enc, w1c, want_stats=training)
        if training:
            st = ops.bn_finalize(part, n, hdim, gamma.detach(), beta.detach(), bn.running_mean, bn.running_var,
                                 bn.num_batches_tracked, b1.detach(), bn.eps, bn.momentum)
        else:
            # eval: BN(h + b1) with running stats == h*scale + (shift + b1*scale)
            ident = ops.BNState(hdim, enc.device)
            ident.scale.fill_(1.0)
            ident.shift.copy_(b1.detach())
            h = ops.bn_apply(h, ident, n, hdim, False)
            st = ops.bn_eval_state(hdim, gamma.detach(), beta.detach(), bn.running_mean, bn.running_var, bn.eps)
        a = ops.bn_apply(h, st, n, hdim, True)
        p, _ = ops.conv2d_fwd(d2, a, w2c, want_stats=False)
        ctx.saved = (enc, h, a, st, d1, d2, w1, w2, gamma, training)
        return p.view(n, odim)

    @staticmethod
    def backward(ctx, dp):
        enc, h, a, st, d1, d2, w1, w2, gamma, training = ctx.saved
        if not training:
            raise NotImplementedError("backward through an eval-mode projection head is not part of the training step")
        n = enc.shape[0]
        hdim = w1.shape[0]
        dp = dp.contiguous()
        dw2 = ops.conv2d_wgrad(d2, a, dp)
        da = ops.conv2d_dgrad(d2, dp, ops.pack_crsk(w2.detach().view(*w2.shape, 1, 1), torch.float32))
        dh, _, dgamma, dbeta = ops.bn_backward(da, a, h, st, gamma.detach(), n, hdim, True, False, mask_from_y=True)
        db1 = ops.colsum(dh, n, hdim)
        dw1 = ops.conv2d_wgrad(d1, enc, dh)
        denc = ops.conv2d_dgrad(d1, dh, ops.pack_crsk(w1.detach().view(*w1.shape, 1, 1), torch.float32))
        ctx.saved = None
        return denc.view(n, -1), None, None, dw1.view_as(w1), db1, dgamma, dbeta, dw2.view_as(w2)


class SimCLR(BaseModel):
    """src/models/unsupervised/simclr_model.py:10-86."""

    unwarps = False   # projection-space un-rotate / un-translate (PeCLR family)
    weighted = False  # joint-distance adaptive weights (the *_W family)

    def __init__(self, config, logger_debug=None, mode: str = "train"):
        super().__init__(config, logger_debug, mode)
        self.save_hyperparameters()
        self.projection_head = self.get_projection_head()
        self.mode = mode
        self.process_group = None  # torch.distributed group for the global negative set (None = default / single)

    def get_projection_head(self) -> nn.Sequential:
        # quirk honoured: for resnet 18/34 the JSON's projection_head_input_dim (2048) is wrong and
        # nothing in the reference fixes it -> follow the encoder's width
        in_dim = getattr(self.encoder, "out_features", self.config.projection_head_input_dim)
        return nn.Sequential(
            nn.Linear(in_dim, self.config.projection_head_hidden_dim, bias=True),
            nn.BatchNorm1d(self.config.projection_head_hidden_dim),
            nn.ReLU(),
            nn.Linear(self.config.projection_head_hidden_dim, self.config.output_dim, bias=False),
        )

    # ---- pieces ---------------------------------------------------------------
    def get_encodings(self, batch_images: Tensor) -> Tensor:
        return self.encoder(batch_images)

    def _head(self, enc: Tensor) -> Tensor:
        ph = self.projection_head
        return _HeadFn.apply(enc, ph, self.training, ph[0].weight, ph[0].bias, ph[1].weight, ph[1].bias, ph[3].weight)

    def get_projection_stats(self, projection: Tensor, name: str) -> dict:
        return mu.projection_stats(projection, name)

    def _projections(self, batch: Dict[str, Tensor]) -> Tensor:
        # reference: torch.cat of the two views (simclr_model.py:30-37); the encoder takes the pair and reads each view in place
        x = (batch["transformed_image1"], batch["transformed_image2"])
        b = x[0].shape[0]
        p = self._head(self.get_encodings(x))
        if not self.unwarps:
            return mu.normalize(p)
        hw = tuple(batch["transformed_image1"].shape[-2:])
        stats = {**self.get_projection_stats(p[:b], "proj1"), **self.get_projection_stats(p[b:], "proj2")}
        self.train_metrics = {**self.train_metrics, **stats}
        jx = jy = ang = None
        if "crop" in self.config.augmentation:
            jx = torch.cat((batch["jitter_x_1"], batch["jitter_x_2"]), dim=0)
            jy = torch.cat((batch["jitter_y_1"], batch["jitter_y_2"]), dim=0)
        if "rotate" in self.config.augmentation:
            ang = torch.cat((batch["angle_1"], batch["angle_2"]), dim=0)
        return mu.transformed_projections(p, jx, jy, ang, hw)

    def get_transformed_projections(self, batch: Dict[str, Tensor]) -> Tuple[Tensor, Tensor]:
        z = self._projections(batch)
        b = z.shape[0] // 2
        return z[:b], z[b:]

    def get_adaptive_weights(self, batch: Dict[str, Tensor], joints_type, weight_type, diff_type):
        """Explicit (B,), (N,N) weights like the reference (simhand_w_model.py:96-120); the training step itself
        uses the fused path and never materialises them."""
        j1, j2 = self._joints(batch)
        if weight_type == "linear":
            if getattr(self.config, "use_pca", False):
                return mu.get_weights_linear_with_pca(mu.apply_pca(j1), mu.apply_pca(j2), diff_type)
            return mu.get_weights_linear(j1, j2, diff_type)
        if getattr(self.config, "use_pca", False):
            return mu.get_weights_nonlinear_with_pca(mu.apply_pca(j1), mu.apply_pca(j2), self.config.non_linear_lambda_pos,
                                                     self.config.non_linear_lambda_neg, diff_type)
        return mu.get_weights_nonlinear(j1, j2, self.config.non_linear_lambda_pos, self.config.non_linear_lambda_neg, diff_type)

    def _joints(self, batch):
        # `joints_type is 'original'` in the reference is an identity test on a literal -> always False for
        # CLI / JSON strings: the augmented joints are always used
        return batch["joints1_aug"][:, :, :2], batch["joints2_aug"][:, :, :2]

    # ---- the step ---------------------------------------------------------------
    def contrastive_step(self, batch: Dict[str, Tensor]) -> Tensor:
        z = self._projections(batch)
        cfg = LossConfig.from_model_config(self.config, self.weighted)
        joints = None
        if self.weighted:
            j1, j2 = self._joints(batch)
            if getattr(self.config, "use_pca", False) and type(self).__name__ == "HandCLR_W":
                from .dist_loss import _world

                if _world(self.process_group)[0] > 1:
                    # the reference's PCA basis comes from a randomised torch.pca_lowrank of the batch it sees
                    # (src/models/utils.py:192-215): per-rank bases would put the gathered joints in incompatible
                    # coordinates, so the global-negative loss is undefined for this flag
                    raise NotImplementedError("--use_pca is a single-process option: its per-batch randomised PCA basis "
                                              "cannot be shared across ranks")
                j1, j2 = mu.apply_pca(j1), mu.apply_pca(j2)
                cfg.diff_type = "l2"
            joints = torch.cat((j1, j2), dim=0).to(torch.float32).reshape(z.shape[0], -1).contiguous()
        loss = mu.weighted_ntxent(z, joints, cfg, self.process_group)
        self.log("contrastive_loss", loss, on_step=True, on_epoch=True, prog_bar=True, logger=True)
        return loss

    def forward(self, x: Tensor) -> Dict[str, Tensor]:
        # the reference runs the encoder twice here (simclr_model.py:65-67); one pass gives the same outputs
        embedding = self.encoder(x)
        return {"embedding": embedding, "projection": self._head(embedding)}

    def training_step(self, batch: dict, batch_idx: int) -> Dict[str, Tensor]:
        loss = self.contrastive_step(batch)
        self.train_metrics = {**self.train_metrics, **{"loss": loss}}
        self.plot_params = {"image1": batch["transformed_image1"], "image2": batch["transformed_image2"],
                            "params": {k: v for k, v in batch.items() if "image" not in k}}
        return self.train_metrics

    def validation_step(self, batch: dict, batch_idx: int) -> Dict[str, Tensor]:
        with torch.no_grad():
            loss = self.contrastive_step(batch)
        self.plot_params = {"image1": batch["transformed_image1"], "image2": batch["transformed_image2"],
                            "params": {k: v for k, v in batch.items() if "image" not in k}}
        return {"loss": loss}


class SimCLR_W(SimCLR):
    """src/models/unsupervised/simclr_w_model.py:22-125."""
    weighted = True


class PeCLR(SimCLR):
    """src/models/unsupervised/peclr_model.py:16-111."""
    unwarps = True


class PeCLR_W(SimCLR):
    """src/models/unsupervised/peclr_w_model.py:20-138."""
    unwarps = True
    weighted = True


class SiMHand(PeCLR):
    """src/models/unsupervised/simhand_model.py:15-107 (same step as PeCLR)."""


class SiMHand_BASE(PeCLR):
    """src/models/unsupervised/simhand_base_model.py:15-116.  (The reference calls translate_encodings /
    rotate_encoding without their `logger` argument there, App. D #5; the arithmetic is PeCLR's.)"""


class HandCLR_W(SimCLR):
    """src/models/unsupervised/simhand_w_model.py:23-151 -- the SiMHand model (README `handclr_w`)."""
    unwarps = True
    weighted = True


class HandCLR_VIS(PeCLR):
    """src/models/unsupervised/simhand_vis_model.py:19-149: PeCLR step + a per-iteration .npy dump of the batch
    for visualisation.  The dump is plotting support and out of scope; the step arithmetic is kept."""


SiMHand_W = HandCLR_W      # names experiments/utils.py:21 imports
SiMHand_VIS = HandCLR_VIS  # names experiments/utils.py:22 imports
