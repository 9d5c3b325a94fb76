"""Functional layer with the reference's names and signatures
(mirror of the hot-path part of src/models/utils.py:157-501,606-684,748-764),
every FLOP in HIP kernels.

Two levels:
  * the reference's explicit surface -- ``vanila_*contrastive_loss``,
    ``get_weights_*``, ``translate_encodings``, ``rotate_encoding`` -- kept so
    code written against src/models/utils.py keeps working (weights are
    materialised as (B,) / (N,N) tensors there, as in the reference);
  * the fused product path used by the step classes --
    ``transformed_projections`` (one kernel for normalize -> translate ->
    rotate -> normalize) and ``weighted_ntxent`` (distances, weights, loss and
    closed-form backward without materialising N x N x 21 x 2, row-block
    sharded over ranks with RCCL; see host/dist_loss.py).
"""
from __future__ import annotations

from typing import Optional, Tuple

import torch
from torch import Tensor

from .. import _lib, ops
from . import dist_loss

TEMPERATURE = 0.5


# --------------------------------------------------------------------------
# fused post-process (a3/a6/a7/a8)
# --------------------------------------------------------------------------
class _PostprocessFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, p, jx, jy, angle, hw, flags, tx, ty):
        p = p.contiguous().float()  # the kernels read fp32 rows of 128 (ops refuses anything else)
        ctx.save_for_backward(p)
        ctx.aux = (jx, jy, angle, hw, flags, tx, ty)
        return ops.proj_postprocess_fwd(p, jx, jy, angle, hw, flags, tx, ty)

    @staticmethod
    def backward(ctx, dz):
        (p,) = ctx.saved_tensors
        jx, jy, angle, hw, flags, tx, ty = ctx.aux
        dp = ops.proj_postprocess_bwd(p, jx, jy, angle, hw, dz.contiguous().float(), flags, tx, ty)
        return dp, None, None, None, None, None, None, None


def transformed_projections(head_out: Tensor, jitter_x: Optional[Tensor], jitter_y: Optional[Tensor],
                            angle: Optional[Tensor], image_hw: Tuple[int, int]) -> Tensor:
    """normalize -> translate (if jitters given) -> rotate (if angles given) ->
    normalize on (N,128) rows; simhand_w_model.py:56-93.  jitter_*: int64 (N,),
    angle: float64 (N,) degrees, as collated."""
    if jitter_x is not None:
        jitter_x, jitter_y = jitter_x.to(torch.int64).contiguous(), jitter_y.to(torch.int64).contiguous()
    if angle is not None:
        angle = angle.to(torch.float64).contiguous()
    return _PostprocessFn.apply(head_out, jitter_x, jitter_y, angle, tuple(image_hw), _lib.PP_FUSED, None, None)


def translate_encodings(encoding: Tensor, translate_x: Tensor, translate_y: Tensor, logger=None) -> Tensor:
    """src/models/utils.py:661-684; encoding (N,64,2).  Returns a new tensor
    (the reference updates in place and returns the same object)."""
    n = encoding.shape[0]
    flat = encoding.reshape(n, -1)
    tx = translate_x.to(torch.float32).contiguous()
    ty = translate_y.to(torch.float32).contiguous()
    return _PostprocessFn.apply(flat, None, None, None, (1, 1), 0, tx, ty).view_as(encoding)


def rotate_encoding(encoding: Tensor, angle: Tensor, logger=None) -> Tensor:
    """src/models/utils.py:636-658; ``angle`` is used as given (the step
    passes ``-angles``)."""
    n = encoding.shape[0]
    flat = encoding.reshape(n, -1)
    ang = angle.to(torch.float64).contiguous()
    return _PostprocessFn.apply(flat, None, None, ang, (1, 1), _lib.PP_ANGLE_AS_GIVEN, None, None).view_as(encoding)


def normalize(x: Tensor) -> Tensor:
    """F.normalize(x) on (N,128) rows."""
    return _PostprocessFn.apply(x, None, None, None, (1, 1), _lib.PP_NORM_OUT, None, None)


def projection_stats(head_out_view: Tensor, name: str) -> dict:
    """get_projection_stats, simhand_w_model.py:138-151, for one view (B,128)
    or (B,64,2) of the raw head output (detached)."""
    flat = head_out_view.detach().reshape(head_out_view.shape[0], -1).contiguous()
    s = ops.proj_stats(flat)
    keys = ("x_mean", "x_median", "x_min", "x_max", "y_mean", "y_median", "y_min", "y_max")
    return {f"{name}{k}": s[i] for i, k in enumerate(keys)}


# --------------------------------------------------------------------------
# adaptive weights, explicit surface (a10)
# --------------------------------------------------------------------------
def _joint_rows(j1: Tensor, j2: Tensor) -> Tuple[Tensor, int, str]:
    b = j1.shape[0]
    j = torch.cat((j1, j2), dim=0).to(torch.float32).reshape(2 * b, -1).contiguous()
    return j, b, ("l2" if j1.dim() == 2 else "")


def _weights(j1, j2, weight_type, diff_type, lam_pos=0.0, lam_neg=0.0):
    j, b, forced = _joint_rows(j1, j2)
    mode = forced or diff_type
    stats = torch.zeros(8, dtype=torch.float64, device=j.device)
    dp = ops.pos_dist(j, b, mode, stats)
    dn = ops.neg_dist(j, b, mode, b, 0, stats)
    wp = ops.weights_from_dist(dp, weight_type, stats, True, b, lam_pos)
    wn = ops.weights_from_dist(dn, weight_type, stats, False, float(4 * b * b), lam_neg)
    return wp, wn


def get_weights_linear(joints1: Tensor, joints2: Tensor, diff_type: str):
    """src/models/utils.py:218-261 -> (pos_weights (B,), neg_weights (N,N))."""
    return _weights(joints1, joints2, "linear", diff_type)


def get_weights_nonlinear(joints1: Tensor, joints2: Tensor, lambda_pos: float, lambda_neg: float, diff_type: str):
    """src/models/utils.py:304-346."""
    return _weights(joints1, joints2, "non_linear", diff_type, lambda_pos, lambda_neg)


def get_weights_linear_with_pca(joints1: Tensor, joints2: Tensor, diff_type: str):
    """src/models/utils.py:264-301 (inputs are (B,14) PCA features)."""
    return _weights(joints1, joints2, "linear", diff_type)


def get_weights_nonlinear_with_pca(joints1: Tensor, joints2: Tensor, lambda_pos: float, lambda_neg: float, diff_type: str):
    """src/models/utils.py:349-388."""
    return _weights(joints1, joints2, "non_linear", diff_type, lambda_pos, lambda_neg)


def apply_pca(joints: Tensor, target_dim: int = 14) -> Tensor:
    """src/models/utils.py:192-215.  As in the reference this runs on the HOST
    (``.cpu()`` + randomised ``torch.pca_lowrank``, :209-213) and is therefore
    only reproducible under the same torch RNG stream; it is not on the
    measured path (``--use_pca`` is off in every README recipe)."""
    if joints.dim() != 3 or tuple(joints.shape[1:]) != (21, 2):
        raise ValueError(f"Expected joints to have shape (batch, 21, 2), but got {joints.shape}")
    flat = joints.contiguous().view(joints.shape[0], -1).float().cpu()
    _, _, v = torch.pca_lowrank(flat, q=target_dim)
    return torch.matmul(flat, v[:, :target_dim]).to(joints.device)


# --------------------------------------------------------------------------
# NT-Xent, explicit surface (a11)
# --------------------------------------------------------------------------
def _explicit_loss(z1, z2, pos_w, neg_w, temperature):
    cfg = dist_loss.LossConfig(weight_type="explicit" if (pos_w is not None or neg_w is not None) else None,
                               use_wpos=pos_w is not None, use_wneg=neg_w is not None, temperature=temperature)
    z = torch.cat((z1, z2), dim=0)
    return dist_loss.ShardedNtxent.apply(z, None, cfg, None, pos_w, neg_w)


def vanila_contrastive_loss(z1: Tensor, z2: Tensor, temperature: float = TEMPERATURE) -> Tensor:
    """src/models/utils.py:157-189."""
    return _explicit_loss(z1, z2, None, None, temperature)


def vanila_weights_contrastive_loss(z1, z2, pos_weights, neg_weights, temperature: float = TEMPERATURE) -> Tensor:
    """src/models/utils.py:391-427."""
    return _explicit_loss(z1, z2, pos_weights, neg_weights, temperature)


def vanila_pos_weights_contrastive_loss(z1, z2, pos_weights, temperature: float = TEMPERATURE) -> Tensor:
    """src/models/utils.py:430-465."""
    return _explicit_loss(z1, z2, pos_weights, None, temperature)


def vanila_neg_weights_contrastive_loss(z1, z2, neg_weights, temperature: float = TEMPERATURE) -> Tensor:
    """src/models/utils.py:468-501."""
    return _explicit_loss(z1, z2, None, neg_weights, temperature)


# --------------------------------------------------------------------------
# fused product path
# --------------------------------------------------------------------------
def weighted_ntxent(z_local: Tensor, joints_local: Optional[Tensor], cfg: "dist_loss.LossConfig", group=None) -> Tensor:
    """Loss over the GATHERED global batch.  z_local (2*b_loc,128) rows
    cat(view1, view2) of this rank's pairs; joints_local (2*b_loc,F) or None."""
    return dist_loss.ShardedNtxent.apply(z_local, joints_local, cfg, group, None, None)


def get_wrapper_model(config, pretrained: bool, wrapper: bool = False, compute_dtype: torch.dtype = torch.float32):
    """src/models/utils.py:748-764."""
    from .config import edict
    from .resnet_model import ResNetModel

    if wrapper:
        raise NameError("WrapperModel is undefined in the reference as well (src/models/utils.py:761-762)")
    cfg = edict({"model": {"backend_model": "resnet" + str(config.resnet_size), "norm_layer": "bn", "use_var": False,
                           "pretrained": pretrained},
                 "dataset": {"np": 21}, "loss": {"hmap": {"enabled": False}}})
    return ResNetModel(config=cfg, mode="pretraining", compute_dtype=compute_dtype)
