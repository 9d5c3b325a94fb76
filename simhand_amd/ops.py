"""Raw (non-autograd) operator layer: torch tensors in, torch tensors out, every
FLOP in ``libsimhand_hip.so``.  torch only owns memory and the stream.

Layouts: activations NHWC, conv weights KRSC rows ``[cout][k_pad]``; ``dtype``
is the compute dtype of activations / packed weights (torch.float32 for the
parity mode, torch.bfloat16 for the MFMA bf16 path).
"""
from __future__ import annotations

import ctypes as C
from typing import Optional, Tuple

import numpy as np
import torch

from . import _lib
from ._lib import BnBwdFuse, ConvDesc, DgradOpts, DySrc, NtxentParams, check

# torch.float16 maps to the same enum as torch.bfloat16: "the 16-bit storage type" -- which of the two it is, is a property of the library
# BUILD (libsimhand_hip.so / libsimhand_hip_f16.so, selected with _lib.use_half): dt() refuses a tensor of the other format
_DT = {torch.float32: _lib.SH_F32, torch.bfloat16: _lib.SH_BF16, torch.float16: _lib.SH_BF16}


def _stream() -> C.c_void_p:
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _ptr(t: Optional[torch.Tensor], dtype: Optional[torch.dtype] = None) -> C.c_void_p:
    """Device pointer of a contiguous tensor.  `dtype`: the element type the C entry point reads the buffer as -- a tensor
    of another dtype would be silently reinterpreted, so it is refused here."""
    if t is None:
        return C.c_void_p(0)
    if not t.is_cuda:
        raise _lib.SimhandHipError("simhand_amd ops need device tensors (there is no CPU path)")
    if not t.is_contiguous():
        raise _lib.SimhandHipError("simhand_amd ops need contiguous tensors")
    if dtype is not None and t.dtype != dtype:
        raise _lib.SimhandHipError(f"simhand_amd op expects a {dtype} tensor, got {t.dtype}")
    return C.c_void_p(t.data_ptr())


_F32, _F64, _I64 = torch.float32, torch.float64, torch.int64
PROJ_DIM = 128  # the post-process / loss kernels view a row as 64 2-D points (output_dim of every *_config.json)


def _require_width(t: torch.Tensor, what: str) -> None:
    if t.dim() != 2 or t.shape[1] != PROJ_DIM:
        raise _lib.SimhandHipError(f"{what}: rows must be {PROJ_DIM} wide (64 2-D points, output_dim of the reference configs); "
                                   f"got shape {tuple(t.shape)}")


def _lib_dev():
    _lib.require_device()
    return _lib.load()


def dt(dtype: torch.dtype) -> int:
    if dtype is not torch.float32 and dtype is not _lib.half_dtype():
        raise _lib.SimhandHipError(f"{dtype} tensors with the {_lib.half_format()} build of the library: call _lib.use_half() / "
                                   f"set_compute_dtype() for the format you store in")
    return _DT[dtype]


def H16() -> torch.dtype:
    """torch dtype of the current build's 16-bit storage format."""
    return _lib.half_dtype()


# --------------------------------------------------------------------------- loss
class NtxentPlan:
    """Shapes + flags of one loss evaluation; owns the workspaces."""

    def __init__(self, B: int, b_loc: int, pair_off: int, weight_type: Optional[str], use_wpos: bool, use_wneg: bool,
                 temperature: float = 0.5, lambda_pos: float = 0.0, lambda_neg: float = 0.0, dim: int = 128):
        self.p = NtxentParams(B, dim, b_loc, pair_off, _lib.WEIGHT_TYPES[weight_type], int(use_wpos), int(use_wneg),
                              temperature, lambda_pos or 0.0, lambda_neg or 0.0)
        self.B, self.N, self.b_loc, self.pair_off, self.rows = B, 2 * B, b_loc, pair_off, 2 * b_loc
        self.weighted = weight_type not in (None, "none")
        self.ws_bytes = _lib.load().simhand_ntxent_workspace_bytes(C.byref(self.p))
        if self.ws_bytes == 0:
            check(1, "simhand_ntxent_workspace_bytes")


def pos_dist(J_all: torch.Tensor, B: int, mode: str, stats: torch.Tensor) -> torch.Tensor:
    lib = _lib_dev()
    d = torch.empty(B, dtype=torch.float32, device=J_all.device)
    check(lib.simhand_pos_dist(_ptr(J_all, _F32), B, J_all.shape[1], _lib.DIST_MODES[mode], _ptr(d), _ptr(stats, _F64), _stream()), "pos_dist")
    return d


def neg_dist(J_all: torch.Tensor, B: int, mode: str, b_loc: int, pair_off: int, stats: torch.Tensor, stats_only: bool = False):
    """D row block [rows_loc][N] + its max / min / sum into stats[0..2]; stats_only: no block is written (returns None) --
    the fused loss kernels recompute the distances in their tiles."""
    lib = _lib_dev()
    rows, N = 2 * b_loc, 2 * B
    D = None if stats_only else torch.empty(rows, N, dtype=torch.float32, device=J_all.device)
    nb = lib.simhand_neg_dist_workspace_bytes(rows, N)
    ws = torch.empty(nb, dtype=torch.uint8, device=J_all.device)
    check(lib.simhand_neg_dist(_ptr(J_all, _F32), B, J_all.shape[1], _lib.DIST_MODES[mode], b_loc, pair_off, _ptr(D), _ptr(stats, _F64),
                               _ptr(ws), nb, _stream()), "neg_dist")
    return D


def weights_from_dist(dist: torch.Tensor, weight_type: str, stats: torch.Tensor, positive: bool, mean_count: float,
                      lam: float = 0.0) -> torch.Tensor:
    lib = _lib_dev()
    w = torch.empty_like(dist)
    check(lib.simhand_weights_from_dist(_ptr(dist, _F32), dist.numel(), _lib.WEIGHT_TYPES[weight_type], _ptr(stats, _F64), int(positive),
                                        float(mean_count), float(lam or 0.0), _ptr(w), _stream()), "weights_from_dist")
    return w


def ntxent_fwd(plan: NtxentPlan, Z_all, D_loc, d_pos, stats) -> Tuple[torch.Tensor, torch.Tensor]:
    lib = _lib_dev()
    dev = Z_all.device
    _require_width(Z_all, "ntxent_fwd")
    neg = torch.empty(plan.rows, dtype=torch.float32, device=dev)
    loss = torch.empty(1, dtype=torch.float32, device=dev)
    ws = torch.empty(plan.ws_bytes, dtype=torch.uint8, device=dev)
    check(lib.simhand_ntxent_fwd(C.byref(plan.p), _ptr(Z_all, _F32), _ptr(D_loc, _F32), _ptr(d_pos, _F32), _ptr(stats, _F64), _ptr(neg), _ptr(loss),
                                 _ptr(ws), plan.ws_bytes, _stream()), "ntxent_fwd")
    return neg, loss


def ntxent_fwd_fused(plan: NtxentPlan, Z_all, J_all, mode: str, d_pos, stats) -> Tuple[torch.Tensor, torch.Tensor]:
    """ntxent_fwd with the joint-distance tile computed inside the loss kernel (no D block in HBM)."""
    lib = _lib_dev()
    dev = Z_all.device
    _require_width(Z_all, "ntxent_fwd_fused")
    neg = torch.empty(plan.rows, dtype=torch.float32, device=dev)
    loss = torch.empty(1, dtype=torch.float32, device=dev)
    ws = torch.empty(plan.ws_bytes, dtype=torch.uint8, device=dev)
    check(lib.simhand_ntxent_fwd_fused(C.byref(plan.p), _ptr(Z_all, _F32), _ptr(J_all, _F32), J_all.shape[1], _lib.DIST_MODES[mode], _ptr(d_pos, _F32),
                                       _ptr(stats, _F64), _ptr(neg), _ptr(loss), _ptr(ws), plan.ws_bytes, _stream()), "ntxent_fwd_fused")
    return neg, loss


def ntxent_bwd_fused(plan: NtxentPlan, Z_all, J_all, mode: str, d_pos, stats, neg_all, dloss) -> torch.Tensor:
    lib = _lib_dev()
    dev = Z_all.device
    _require_width(Z_all, "ntxent_bwd_fused")
    dZ = torch.empty(plan.rows, plan.p.dim, dtype=torch.float32, device=dev)
    ws = torch.empty(plan.ws_bytes, dtype=torch.uint8, device=dev)
    check(lib.simhand_ntxent_bwd_fused(C.byref(plan.p), _ptr(Z_all, _F32), _ptr(J_all, _F32), J_all.shape[1], _lib.DIST_MODES[mode], _ptr(d_pos, _F32),
                                       _ptr(stats, _F64), _ptr(neg_all, _F32), _ptr(dloss, _F32), _ptr(dZ), _ptr(ws), plan.ws_bytes, _stream()),
          "ntxent_bwd_fused")
    return dZ


def ntxent_bwd(plan: NtxentPlan, Z_all, D_loc, d_pos, stats, neg_all, dloss) -> torch.Tensor:
    lib = _lib_dev()
    dev = Z_all.device
    _require_width(Z_all, "ntxent_bwd")
    dZ = torch.empty(plan.rows, plan.p.dim, dtype=torch.float32, device=dev)
    ws = torch.empty(plan.ws_bytes, dtype=torch.uint8, device=dev)
    check(lib.simhand_ntxent_bwd(C.byref(plan.p), _ptr(Z_all, _F32), _ptr(D_loc, _F32), _ptr(d_pos, _F32), _ptr(stats, _F64), _ptr(neg_all, _F32), _ptr(dloss, _F32),
                                 _ptr(dZ), _ptr(ws), plan.ws_bytes, _stream()), "ntxent_bwd")
    return dZ


# ------------------------------------------------------------------ post-process
def proj_postprocess_fwd(P, jx, jy, angle, hw, flags: int = _lib.PP_FUSED, tx=None, ty=None) -> torch.Tensor:
    lib = _lib_dev()
    _require_width(P, "proj_postprocess_fwd")
    Z = torch.empty_like(P)
    check(lib.simhand_proj_postprocess_fwd(_ptr(P, _F32), P.shape[0], P.shape[1], _ptr(jx, _I64), _ptr(jy, _I64), _ptr(tx, _F32), _ptr(ty, _F32), _ptr(angle, _F64), int(hw[0]),
                                           int(hw[1]), flags, _ptr(Z), _stream()), "proj_postprocess_fwd")
    return Z


def proj_postprocess_bwd(P, jx, jy, angle, hw, dZ, flags: int = _lib.PP_FUSED, tx=None, ty=None) -> torch.Tensor:
    lib = _lib_dev()
    _require_width(P, "proj_postprocess_bwd")
    _require_width(dZ, "proj_postprocess_bwd (dZ)")
    dP = torch.empty_like(P)
    check(lib.simhand_proj_postprocess_bwd(_ptr(P, _F32), P.shape[0], P.shape[1], _ptr(jx, _I64), _ptr(jy, _I64), _ptr(tx, _F32), _ptr(ty, _F32), _ptr(angle, _F64), int(hw[0]),
                                           int(hw[1]), flags, _ptr(dZ, _F32), _ptr(dP), _stream()), "proj_postprocess_bwd")
    return dP


def proj_stats(P: torch.Tensor) -> torch.Tensor:
    """8 batch-mean statistics of one view's raw head output (rows x 128)."""
    lib = _lib_dev()
    _require_width(P, "proj_stats")
    n = P.shape[0]
    ws = torch.empty(n, 8, dtype=torch.float32, device=P.device)
    out = torch.empty(8, dtype=torch.float32, device=P.device)
    check(lib.simhand_proj_stats(_ptr(P, _F32), n, P.shape[1], _ptr(ws), _ptr(out), _stream()), "proj_stats")
    return out


# ------------------------------------------------------------------------ conv
def conv_desc(n, h, w, cin, cout, r, s, stride, pad, dtype: torch.dtype) -> ConvDesc:
    ho = (h + 2 * pad - r) // stride + 1
    wo = (w + 2 * pad - s) // stride + 1
    return ConvDesc(n, h, w, cin, cout, r, s, stride, pad, ho, wo, dt(dtype))


def conv2d_fwd(d: ConvDesc, x, w, want_stats: bool = True):
    lib = _lib_dev()
    y = torch.empty(d.n, d.ho, d.wo, d.cout, dtype=x.dtype, device=x.device)
    part = None
    if want_stats:
        nblk = lib.simhand_conv2d_fwd_stat_blocks(C.byref(d))
        part = torch.empty(nblk, 2, d.cout, dtype=torch.float32, device=x.device)
    check(lib.simhand_conv2d_fwd(C.byref(d), _ptr(x), _ptr(w), _ptr(y), _ptr(part), _stream()), "conv2d_fwd")
    return y, part


def conv2d_fwd_bnin_ok(d: ConvDesc) -> bool:
    return bool(_lib_dev().simhand_conv2d_fwd_bnin_ok(C.byref(d)))


def conv2d_fwd_bnin(d: ConvDesc, y_in, st_in: "BNState", w, want_stats: bool = True):
    """a = relu(y_in * st_in.scale + st_in.shift); y = conv(a, w) in one launch: the BatchNorm + ReLU of the unit in front is applied inside
    the 3x3 kernel's LDS ring and a leaves as a by-product.  Returns (a, y, partial sums or None); bit-identical to bn_apply + conv2d_fwd."""
    lib = _lib_dev()
    a = torch.empty_like(y_in)
    y = torch.empty(d.n, d.ho, d.wo, d.cout, dtype=y_in.dtype, device=y_in.device)
    part = None
    if want_stats:
        nblk = lib.simhand_conv2d_fwd_stat_blocks(C.byref(d))
        part = torch.empty(nblk, 2, d.cout, dtype=torch.float32, device=y_in.device)
    check(lib.simhand_conv2d_fwd_bnin(C.byref(d), _ptr(y_in), _ptr(st_in.scale), _ptr(st_in.shift), _ptr(w), _ptr(a), _ptr(y), _ptr(part),
                                      _stream()), "conv2d_fwd_bnin")
    return a, y, part


def conv2d_fwd_bnact(d: ConvDesc, x, w, st: "BNState", relu: bool, residual=None, want_mask: bool = False):
    """out = act(conv(x) * st.scale + st.shift (+ residual)) with the BN / residual / ReLU tail in the conv epilogue
    (bf16; the raw conv output is never stored).  Returns out or (out, ReLU bit mask)."""
    lib = _lib_dev()
    out = torch.empty(d.n, d.ho, d.wo, d.cout, dtype=x.dtype, device=x.device)
    mask = torch.empty(d.n * d.ho * d.wo, d.cout // 8, dtype=torch.uint8, device=x.device) if want_mask else None
    check(lib.simhand_conv2d_fwd_bnact(C.byref(d), _ptr(x), _ptr(w), _ptr(st.scale), _ptr(st.shift), _ptr(residual), int(relu), _ptr(out),
                                       _ptr(mask), _stream()), "conv2d_fwd_bnact")
    return (out, mask) if want_mask else out


def conv2d_fwd_chain_ok(d: ConvDesc) -> bool:
    return bool(_lib_dev().simhand_conv2d_fwd_chain_ok(C.byref(d)))


def conv2d_fwd_bnact_chain(d: ConvDesc, x, w, st: "BNState", residual, chain_w):
    """conv2d_fwd_bnact (residual + ReLU + bit mask) with the next block's conv1 chained on (simhand_conv2d_fwd_bnact_chain):
    returns (out, mask, chain_y [n][ho][wo][cin] raw conv output of that conv1, its BatchNorm partial sums)."""
    lib = _lib_dev()
    m = d.n * d.ho * d.wo
    out = torch.empty(d.n, d.ho, d.wo, d.cout, dtype=x.dtype, device=x.device)
    mask = torch.empty(m, d.cout // 8, dtype=torch.uint8, device=x.device)
    cy = torch.empty(d.n, d.ho, d.wo, d.cin, dtype=x.dtype, device=x.device)
    part = torch.empty(lib.simhand_conv2d_fwd_chain_stat_blocks(C.byref(d)), 2, d.cin, dtype=torch.float32, device=x.device)
    check(lib.simhand_conv2d_fwd_bnact_chain(C.byref(d), _ptr(x, H16()), _ptr(w, H16()), _ptr(st.scale), _ptr(st.shift),
                                             _ptr(residual, H16()), _ptr(out), _ptr(mask), _ptr(chain_w, H16()), _ptr(cy),
                                             _ptr(part), _stream()), "conv2d_fwd_bnact_chain")
    return out, mask, cy, part


def conv2d_dgrad(d: ConvDesc, dy, wt, dx: Optional[torch.Tensor] = None, accumulate: bool = False):
    lib = _lib_dev()
    if dx is None:
        dx = torch.empty(d.n, d.h, d.w, d.cin, dtype=dy.dtype, device=dy.device)
    check(lib.simhand_conv2d_dgrad(C.byref(d), _ptr(dy), _ptr(wt), _ptr(dx), int(accumulate), _stream()), "conv2d_dgrad")
    return dx


def conv2d_dgrad_masked_residual(d: ConvDesc, dy, wt, res_grad, res_mask) -> torch.Tensor:
    """dx = dgrad(dy) + res_grad * relu-bit(res_mask): identity-block merge without a materialised residual gradient."""
    lib = _lib_dev()
    dx = torch.empty(d.n, d.h, d.w, d.cin, dtype=dy.dtype, device=dy.device)
    check(lib.simhand_conv2d_dgrad_masked_residual(C.byref(d), _ptr(dy), _ptr(wt), _ptr(dx), _ptr(res_grad), _ptr(res_mask), _stream()),
          "conv2d_dgrad_masked_residual")
    return dx


def conv2d_dgrad_fused(d: ConvDesc, dy, wt, prev_y, prev_st: Optional["BNState"], prev_mask, dx: Optional[torch.Tensor] = None,
                       accumulate: bool = False, res_grad=None, res_mask=None):
    """Data gradient whose epilogue also emits the BatchNorm-backward partial sums (sum g, sum g*y) of the PREVIOUS
    conv+BN unit -- the one whose incoming gradient is the dx produced here (prev_y = its raw conv output).
    prev_mask given: residual unit, ReLU bit mask; else prev_st given: ReLU mask recomputed from prev_y; neither: no
    ReLU.  Returns (dx, raw_partial) for bn_backward(raw_partial=...)."""
    lib = _lib_dev()
    if dx is None:
        dx = torch.empty(d.n, d.h, d.w, d.cin, dtype=dy.dtype, device=dy.device)
    mode = 2 if res_grad is not None else int(accumulate)
    relu_mode = 3 if prev_mask is not None else (2 if prev_st is not None else 0)
    nblk = lib.simhand_conv2d_dgrad_stat_blocks(C.byref(d), mode, relu_mode, 0)
    part = torch.empty(nblk, 2, d.cin, dtype=torch.float32, device=dy.device)
    f = BnBwdFuse()
    f.y = _ptr(prev_y)
    f.mask = _ptr(prev_mask)
    f.relu_mode = relu_mode
    f.scale = _ptr(prev_st.scale) if f.relu_mode == 2 else None
    f.shift = _ptr(prev_st.shift) if f.relu_mode == 2 else None
    f.partial = _ptr(part)
    check(lib.simhand_conv2d_dgrad_fused(C.byref(d), _ptr(dy), _ptr(wt), _ptr(dx), mode, _ptr(res_grad), _ptr(res_mask), C.byref(f),
                                         _stream()), "conv2d_dgrad_fused")
    return dx, part


def conv2d_dgrad_ex(d: ConvDesc, dy, wt, dx: Optional[torch.Tensor] = None, accumulate: bool = False, res_grad=None, res_mask=None,
                    bias=None, fuse_mode: Optional[int] = None, prev_y=None, prev_st: Optional["BNState"] = None, prev_mask=None,
                    want_sums: bool = True, x2=None, wt2=None, dy_src=None, fp8=None, sub_grad=None):
    """General data gradient (simhand_conv2d_dgrad_ex): optional accumulate / masked-residual merge, fp32 per-channel
    bias, and epilogue fusion.  fuse_mode: None = none; 0 / 2 / 3 = BN-backward sums of the previous unit (no ReLU /
    mask from prev_y*scale+shift / bit mask); 4 = store the gradient masked by prev_mask and emit its channel sums.
    x2 / wt2: second reduction segment, dx = dy wt^T + x2 wt2^T in one pass (only where conv2d_dgrad_concat_ok).
    fp8 = (dy_q, wt_q, dy scaler, weight scaler): the reduction runs over e4m3 operands (only where conv2d_dgrad_fp8_pays).
    dy_src = (da, y, BNState, (coef_a, coef_b, coef_c), relu, dy_out): the dy operand is derived on load as the BatchNorm-backward
    apply of the unit (sh_dy_src; `dy` is ignored, pass None) and written to dy_out; only where conv2d_dgrad_dysrc_ok.
    sub_grad [n][h/2][w/2][cin]: added at the even pixels of dx before the gate (a stride-2 shortcut's dense data gradient; accumulate off).
    Returns (dx, partial or None)."""
    lib = _lib_dev()
    ref = dy if dy is not None else dy_src[0]
    if dx is None:
        dx = torch.empty(d.n, d.h, d.w, d.cin, dtype=ref.dtype, device=ref.device)
    o = DgradOpts()
    if dy_src is not None:
        da, ysrc, st_, coefs, relu_, dy_out = dy_src
        sdy = DySrc()
        sdy.da, sdy.y, sdy.scale, sdy.shift = _ptr(da, H16()), _ptr(ysrc, H16()), _ptr(st_.scale), _ptr(st_.shift)
        sdy.coef_a, sdy.coef_b, sdy.coef_c, sdy.relu, sdy.dy_out = _ptr(coefs[0]), _ptr(coefs[1]), _ptr(coefs[2]), int(relu_), _ptr(dy_out, H16())
        o.dy_src = C.pointer(sdy)
        dy = da
    o.accumulate = 2 if res_grad is not None else int(accumulate)
    o.res_grad = _ptr(res_grad)
    o.res_mask = _ptr(res_mask)
    o.bias = _ptr(bias)
    if sub_grad is not None:
        assert not accumulate and res_grad is None and tuple(sub_grad.shape) == (d.n, d.h // 2, d.w // 2, d.cin) and d.h % 2 == 0 and d.w % 2 == 0
        o.sub_grad = _ptr(sub_grad, ref.dtype)
    if x2 is not None:
        o.x2, o.wt2, o.c2 = _ptr(x2), _ptr(wt2), x2.shape[-1]
    if fp8 is not None:
        o.dy_q, o.wt_q = _ptr(fp8[0], torch.uint8), _ptr(fp8[1], torch.uint8)
        o.dy_state, o.w_state = _ptr(fp8[2].state), _ptr(fp8[3].state)
    part = None
    if fuse_mode is not None:
        if want_sums or fuse_mode != 4:  # mode 4 may store the masked gradient without emitting its sums
            nblk = lib.simhand_conv2d_dgrad_stat_blocks(C.byref(d), o.accumulate, fuse_mode, int(o.c2))
            part = torch.empty(nblk, 2, d.cin, dtype=torch.float32, device=ref.device)
        f = BnBwdFuse()
        f.y = _ptr(prev_y)
        f.mask = _ptr(prev_mask)
        f.relu_mode = fuse_mode
        f.scale = _ptr(prev_st.scale) if fuse_mode == 2 else None
        f.shift = _ptr(prev_st.shift) if fuse_mode == 2 else None
        f.partial = _ptr(part)
        o.fuse = C.pointer(f)
    check(lib.simhand_conv2d_dgrad_ex(C.byref(d), _ptr(dy), _ptr(wt), _ptr(dx), C.byref(o), _stream()), "conv2d_dgrad_ex")
    return dx, part


def conv2d_dgrad_dysrc_ok(d: ConvDesc) -> bool:
    return bool(_lib_dev().simhand_conv2d_dgrad_dysrc_ok(C.byref(d)))


def conv2d_dgrad_concat_ok(d: ConvDesc, c2: int) -> bool:
    return bool(_lib_dev().simhand_conv2d_dgrad_concat_ok(C.byref(d), c2))


def conv2d_dgrad_fuse_pays(d: ConvDesc) -> bool:
    return bool(_lib_dev().simhand_conv2d_dgrad_fuse_pays(C.byref(d)))


def conv2d_wgrad(d: ConvDesc, x, dy) -> torch.Tensor:
    """fp32 gradient in KRSC row order [cout][r*s*cin]."""
    lib = _lib_dev()
    nb = lib.simhand_conv2d_wgrad_workspace_bytes(C.byref(d))
    ws = torch.empty(nb, dtype=torch.uint8, device=x.device)
    dw = torch.empty(d.cout, d.r * d.s * d.cin, dtype=torch.float32, device=x.device)
    check(lib.simhand_conv2d_wgrad(C.byref(d), _ptr(x), _ptr(dy), _ptr(dw), _ptr(ws), nb, _stream()), "conv2d_wgrad")
    return dw


def conv2d_wgrad_colsum(d: ConvDesc, x, dy):
    """1x1 / stride-1 bf16: (fp32 KRSC gradient [cout][cin], per-channel sums of dy [cout] fp32) -- the sums ride along
    with the dy tiles the kernel stages anyway."""
    lib = _lib_dev()
    nb = lib.simhand_conv2d_wgrad_workspace_bytes(C.byref(d))
    ws = torch.empty(nb, dtype=torch.uint8, device=x.device)
    dw = torch.empty(d.cout, d.r * d.s * d.cin, dtype=torch.float32, device=x.device)
    splits = lib.simhand_conv2d_wgrad_splits(C.byref(d))
    part = torch.empty(splits, 2, d.cout, dtype=torch.float32, device=x.device)
    if _FOLD_LEGACY or splits >= 4096:  # round-5 form: the channel sums folded by a launch of their own
        check(lib.simhand_conv2d_wgrad_colsum(C.byref(d), _ptr(x), _ptr(dy), _ptr(dw), _ptr(part), _ptr(ws), nb, _stream()), "conv2d_wgrad_colsum")
        return dw, bn_channel_sums(part, d.cout)
    sums = torch.empty(d.cout, dtype=torch.float32, device=x.device)
    check(lib.simhand_conv2d_wgrad_colsum_sums(C.byref(d), _ptr(x), _ptr(dy), _ptr(dw), _ptr(part), _ptr(sums), _ptr(ws), nb, _stream()),
          "conv2d_wgrad_colsum_sums")
    return dw, sums


def bn_apply_gram(y: torch.Tensor, st: "BNState", relu: bool = True):
    """a = act(y*scale + shift), s2 = a^T a (fp32 [c][c]) and sum a ([c]) in ONE launch of the 1x1 weight-gradient kernel
    (bf16; the BatchNorm-apply runs in its operand loader).  y: [n][h][w][c] raw conv output."""
    lib = _lib_dev()
    n, h, w, c = y.shape
    d = conv_desc(n, h, w, c, c, 1, 1, 1, 0, y.dtype)
    nb = lib.simhand_conv2d_wgrad_workspace_bytes(C.byref(d))
    ws = torch.empty(nb, dtype=torch.uint8, device=y.device)
    a = torch.empty_like(y)
    s2 = torch.empty(c, c, dtype=torch.float32, device=y.device)
    splits = lib.simhand_conv2d_wgrad_splits(C.byref(d))
    part = torch.empty(splits, 2, c, dtype=torch.float32, device=y.device)
    if _FOLD_LEGACY or splits >= 4096:
        check(lib.simhand_bn_apply_gram(C.byref(d), _ptr(y, H16()), _ptr(st.scale), _ptr(st.shift), int(relu), _ptr(a), _ptr(s2), _ptr(part),
                                        _ptr(ws), nb, _stream()), "bn_apply_gram")
        return a, s2, bn_channel_sums(part, c)
    t2 = torch.empty(c, dtype=torch.float32, device=y.device)
    check(lib.simhand_bn_apply_gram_sums(C.byref(d), _ptr(y, H16()), _ptr(st.scale), _ptr(st.shift), int(relu), _ptr(a), _ptr(s2), _ptr(part),
                                         _ptr(t2), _ptr(ws), nb, _stream()), "bn_apply_gram_sums")
    return a, s2, t2


def bn_bwd_coefs(st: "BNState", gamma, dgamma, dbeta, m: int):
    """(A, B, C) with dy = A * g - B * y + C: the BatchNorm backward once its sums are reduced (simhand_bn_bwd_coefs)."""
    lib = _lib_dev()
    c = gamma.numel()
    buf = torch.empty(3, c, dtype=torch.float32, device=gamma.device)
    check(lib.simhand_bn_bwd_coefs(_ptr(st.mean), _ptr(st.invstd), _ptr(gamma), _ptr(dgamma), _ptr(dbeta), m, c, _ptr(buf[0]), _ptr(buf[1]),
                                   _ptr(buf[2]), _stream()), "bn_bwd_coefs")
    return buf[0], buf[1], buf[2]


def conv2d_wgrad_bnbwd(d: ConvDesc, x, da, y, st: "BNState", coefs, relu: bool, shape):
    """Weight gradient (OIHW fp32, `shape`) of a 1x1 / stride-1 conv whose output y feeds a BN (+ReLU), with the BatchNorm
    backward apply fused into the dy loader; also returns dy (written once, for the data gradient)."""
    lib = _lib_dev()
    nb = lib.simhand_conv2d_wgrad_workspace_bytes(C.byref(d))
    ws = torch.empty(nb, dtype=torch.uint8, device=x.device)
    dw = torch.empty(shape, dtype=torch.float32, device=x.device)
    dy = torch.empty_like(y)
    c_real = dw[0].numel() // (d.r * d.s)
    check(lib.simhand_conv2d_wgrad_bnbwd(C.byref(d), _ptr(x), _ptr(da), _ptr(y), _ptr(st.scale), _ptr(st.shift), _ptr(coefs[0]), _ptr(coefs[1]),
                                         _ptr(coefs[2]), int(relu), _ptr(dy), _ptr(dw), c_real, _ptr(ws), nb, _stream()), "conv2d_wgrad_bnbwd")
    return dw, dy


def conv2d_wgrad_oihw(d: ConvDesc, x, dy, shape) -> torch.Tensor:
    """fp32 gradient directly in the nn.Conv2d.weight.grad layout `shape` = (cout, c, r, s); for the im2col'd stem
    `d` is the 1x1 descriptor over the padded columns and shape = (64, 3, 7, 7) (147 real columns)."""
    lib = _lib_dev()
    nb = lib.simhand_conv2d_wgrad_workspace_bytes(C.byref(d))
    ws = torch.empty(nb, dtype=torch.uint8, device=x.device)
    dw = torch.empty(shape, dtype=torch.float32, device=x.device)
    c_real = dw[0].numel() // (d.r * d.s)
    check(lib.simhand_conv2d_wgrad_oihw(C.byref(d), _ptr(x), _ptr(dy), _ptr(dw), c_real, _ptr(ws), nb, _stream()), "conv2d_wgrad_oihw")
    return dw


def pack_krsc(w_oihw: torch.Tensor, dtype: torch.dtype, k_pad: Optional[int] = None) -> torch.Tensor:
    lib = _lib_dev()
    k, c, r, s = w_oihw.shape
    k_pad = k_pad or c * r * s
    out = torch.empty(k, k_pad, dtype=dtype, device=w_oihw.device)
    check(lib.simhand_oihw_f32_to_krsc(_ptr(w_oihw), _ptr(out), k, c, r, s, k_pad, dt(dtype), _stream()), "oihw_to_krsc")
    return out


def pack_crsk(w_oihw: torch.Tensor, dtype: torch.dtype) -> torch.Tensor:
    lib = _lib_dev()
    k, c, r, s = w_oihw.shape
    out = torch.empty(c, r * s * k, dtype=dtype, device=w_oihw.device)
    check(lib.simhand_oihw_f32_to_crsk(_ptr(w_oihw), _ptr(out), k, c, r, s, dt(dtype), _stream()), "oihw_to_crsk")
    return out


PACK_ITEM_DTYPE = np.dtype([("src", "<u8"), ("krsc", "<u8"), ("crsk", "<u8"), ("k", "<i4"), ("c", "<i4"), ("r", "<i4"), ("s", "<i4")])  # == sh_pack_item


class PackPlan:
    """Device tables for ``pack_weights_multi``: entries = [(fp32 OIHW master, KRSC buffer, CRSK buffer or None)]; the tables hold raw
    pointers, so a plan is valid as long as those tensors are (the caller keys its cache on their data_ptr()s)."""

    def __init__(self, entries, dtype: torch.dtype):
        lib = _lib_dev()
        ch = lib.simhand_pack_chunk_elems()
        rec = np.zeros(len(entries), dtype=PACK_ITEM_DTYPE)
        pairs, tiles = [], []
        for i, (w, krsc, crsk) in enumerate(entries):
            k, c, r, s = w.shape
            assert w.dtype == torch.float32 and w.is_contiguous() and krsc.dtype == dtype and krsc.numel() == w.numel()
            assert crsk is None or (crsk.dtype == dtype and crsk.numel() == w.numel() and k % 64 == 0 and c % 64 == 0)
            rec[i] = (w.data_ptr(), krsc.data_ptr(), 0 if crsk is None else crsk.data_ptr(), k, c, r, s)
            pairs.extend((i, j) for j in range((w.numel() + ch - 1) // ch))
            if crsk is not None:  # 64 x 64 tile transposes of the KRSC copy, one per (tap, k tile, c tile)
                tiles.extend((i, j) for j in range(r * s * (k // 64) * (c // 64)))
        dev = entries[0][0].device
        self.dtype = dtype
        self.n_chunks, self.n_tiles = len(pairs), len(tiles)
        self.items = torch.from_numpy(rec.view(np.uint8).copy()).to(dev)
        self.chunks = torch.tensor(pairs, dtype=torch.int32).reshape(-1, 2).contiguous().to(dev)
        self.tiles = torch.tensor(tiles or [(0, 0)], dtype=torch.int32).reshape(-1, 2).contiguous().to(dev)


def pack_weights_multi(plan: PackPlan) -> None:
    check(_lib_dev().simhand_pack_weights_multi(_ptr(plan.items), _ptr(plan.chunks), plan.n_chunks, _ptr(plan.tiles), plan.n_tiles,
                                                dt(plan.dtype), _stream()), "pack_weights_multi")


def unpack_krsc_grad(dw: torch.Tensor, shape, k_pad: Optional[int] = None) -> torch.Tensor:
    lib = _lib_dev()
    k, c, r, s = shape
    k_pad = k_pad or c * r * s
    out = torch.empty(k, c, r, s, dtype=torch.float32, device=dw.device)
    check(lib.simhand_krsc_f32_to_oihw(_ptr(dw), _ptr(out), k, c, r, s, k_pad, _stream()), "krsc_to_oihw")
    return out


def cast(x: torch.Tensor, dtype: torch.dtype) -> torch.Tensor:
    lib = _lib_dev()
    out = torch.empty(x.shape, dtype=dtype, device=x.device)
    check(lib.simhand_cast(_ptr(x), dt(x.dtype), _ptr(out), dt(dtype), x.numel(), _stream()), "cast")
    return out


def im2col_nchw(x: torch.Tensor, r, s, stride, pad, k_pad, dtype) -> torch.Tensor:
    lib = _lib_dev()
    n, c, h, w = x.shape
    ho = (h + 2 * pad - r) // stride + 1
    wo = (w + 2 * pad - s) // stride + 1
    col = torch.empty(n, ho, wo, k_pad, dtype=dtype, device=x.device)
    check(lib.simhand_im2col_nchw_f32(_ptr(x), _ptr(col), n, c, h, w, r, s, stride, pad, k_pad, dt(dtype), _stream()), "im2col")
    return col


# ---- direct 7x7/2 stem (no im2col matrix) ----------------------------------------------------------------------------
def stem_geometry(h: int, w: int):
    """(hp, wp, ho, wo) of the zero-padded NHWC4 stem input and the conv1 output."""
    lib = _lib.load()
    v = [C.c_int() for _ in range(4)]
    check(lib.simhand_stem_geometry(h, w, *[C.byref(i) for i in v]), "stem_geometry")
    return tuple(i.value for i in v)


def stem_pad_input(x, dtype) -> torch.Tensor:
    """NCHW fp32 [n,3,h,w] -> zero-padded NHWC4 [n,hp,wp,4] in the compute dtype.  x may be a tuple of such tensors
    (the two views of a contrastive batch): they are packed back to back, i.e. the result equals that of their
    concatenation without the concatenated copy ever being made."""
    lib = _lib_dev()
    views = tuple(x) if isinstance(x, (tuple, list)) else (x,)
    _, c, h, w = views[0].shape
    hp, wp, _, _ = stem_geometry(h, w)
    n = sum(v.shape[0] for v in views)
    xp = torch.empty(n, hp, wp, 4, dtype=dtype, device=views[0].device)
    at = 0
    for v in views:
        assert v.shape[1:] == (3, h, w) and v.dtype == torch.float32 and v.is_contiguous()
        check(lib.simhand_stem_pad_input(_ptr(v), _ptr(xp[at:]), v.shape[0], h, w, dt(dtype), _stream()), "stem_pad_input")
        at += v.shape[0]
    return xp


def stem_pack_weights(w_oihw: torch.Tensor, dtype) -> torch.Tensor:
    lib = _lib_dev()
    assert tuple(w_oihw.shape) == (64, 3, 7, 7) and w_oihw.dtype == torch.float32
    wp = torch.empty(64, 256, dtype=dtype, device=w_oihw.device)
    check(lib.simhand_stem_pack_weights(_ptr(w_oihw.contiguous()), _ptr(wp), dt(dtype), _stream()), "stem_pack_weights")
    return wp


def stem_conv_fwd(xp: torch.Tensor, wp: torch.Tensor, h: int, w: int, want_stats: bool = True):
    lib = _lib_dev()
    n = xp.shape[0]
    _, _, ho, wo = stem_geometry(h, w)
    y = torch.empty(n, ho, wo, 64, dtype=xp.dtype, device=xp.device)
    nblk = lib.simhand_stem_conv_fwd_stat_blocks(n, h, w, dt(xp.dtype))
    part = torch.empty(nblk, 2, 64, dtype=torch.float32, device=xp.device) if want_stats else None
    check(lib.simhand_stem_conv_fwd(_ptr(xp), _ptr(wp), _ptr(y), _ptr(part), n, h, w, dt(xp.dtype), _stream()), "stem_conv_fwd")
    return y, part


def stem_conv_wgrad(xp: torch.Tensor, dy: torch.Tensor, h: int, w: int) -> torch.Tensor:
    """fp32 weight.grad [64,3,7,7] of the stem from the padded input and dy [n,ho,wo,64]."""
    lib = _lib_dev()
    n = xp.shape[0]
    nb = lib.simhand_stem_conv_wgrad_workspace_bytes(n, h, w, dt(xp.dtype))
    ws = torch.empty(nb, dtype=torch.uint8, device=xp.device)
    dw = torch.empty(64, 3, 7, 7, dtype=torch.float32, device=xp.device)
    check(lib.simhand_stem_conv_wgrad(_ptr(xp), _ptr(dy), _ptr(dw), _ptr(ws), nb, n, h, w, dt(xp.dtype), _stream()), "stem_conv_wgrad")
    return dw


def nchw_to_nhwc(x: torch.Tensor, dtype, c_pad: Optional[int] = None) -> torch.Tensor:
    lib = _lib_dev()
    n, c, h, w = x.shape
    c_pad = c_pad or c
    out = torch.empty(n, h, w, c_pad, dtype=dtype, device=x.device)
    check(lib.simhand_nchw_f32_to_nhwc(_ptr(x), _ptr(out), n, c, h, w, c_pad, dt(dtype), _stream()), "nchw_to_nhwc")
    return out


# -------------------------------------------------------------------------- BN
class BNState:
    """Per-call statistics of one train-mode BatchNorm (all fp32, length C).  m_total: under bn_sync, the number of positions
    the statistics were taken over on ALL ranks (None = the local count the caller passes)."""

    __slots__ = ("mean", "invstd", "scale", "shift", "m_total")

    def __init__(self, c: int, device):
        buf = torch.empty(4, c, dtype=torch.float32, device=device)
        self.mean, self.invstd, self.scale, self.shift = buf[0], buf[1], buf[2], buf[3]
        self.m_total = None


# ---- synchronised BatchNorm (optional): statistics over the GLOBAL batch ---------------------------------------------
# Train-mode BatchNorm reduces over positions twice -- forward (sum y, sum y^2) and backward (sum g, sum g * yhat) -- and both
# reductions end in a finalize step that takes per-block partial sums.  Under bn_sync the locally folded sums go through one
# all-reduce (2 C + 1 doubles forward: the position count rides along, so ragged shards are fine; 2 C floats backward) before
# that step: every rank then normalises with the same mean / variance and back-propagates through them.  dgamma / dbeta stay the
# LOCAL sums (they are parameter gradients: the gradient all-reduce adds them up like every other parameter's).
# The reference has no synchronised BatchNorm (its DP replicas keep per-replica statistics, src/experiments/main.py:152-155).
_BN_SYNC = None  # callable(tensor) -> None: in-place SUM all-reduce over the data-parallel group


def set_bn_sync(all_reduce_sum=None) -> None:
    """all_reduce_sum(t): in-place SUM all-reduce of a contiguous fp32 / fp64 device tensor over the ranks whose batches form one
    BatchNorm batch; None switches synchronisation off (per-rank statistics: the default, and what the reference's replicas do)."""
    global _BN_SYNC
    _BN_SYNC = all_reduce_sum


def bn_sync_active() -> bool:
    return _BN_SYNC is not None


_TICKETS: dict = {}


def _ticket(device) -> torch.Tensor:
    """The zero-initialised ticket word of the CURRENT stream on `device` (simhand_bn_finalize_ticket: launches that may run concurrently
    must not share one; every launch leaves it zero)."""
    key = (torch.device(device).index, torch.cuda.current_stream(device).cuda_stream)
    t = _TICKETS.get(key)
    if t is None:
        t = _TICKETS[key] = torch.zeros(16, dtype=torch.int32, device=device)
    return t


def bn_partial_stats(y: torch.Tensor, m: int, c: int) -> torch.Tensor:
    lib = _lib_dev()
    nblk = lib.simhand_bn_stat_blocks(m, c)
    part = torch.empty(nblk, 2, c, dtype=torch.float32, device=y.device)
    check(lib.simhand_bn_partial_stats(_ptr(y), m, c, dt(y.dtype), _ptr(part), _stream()), "bn_partial_stats")
    return part


def bn_finalize(part: torch.Tensor, m: int, c: int, gamma, beta, running_mean, running_var, nbt, pre_bias=None,
                eps: float = 1e-5, momentum: float = 0.1) -> BNState:
    lib = _lib_dev()
    st = BNState(c, part.device)
    if _BN_SYNC is not None:
        # fold the local per-block sums (double, as the finalize kernel does), add the position count, all-reduce, finalize one row
        tot = torch.empty(2 * c + 1, dtype=torch.float64, device=part.device)
        tot[:2 * c] = part.to(torch.float64).sum(0).view(-1)
        tot[2 * c] = float(m)
        _BN_SYNC(tot)
        m = int(round(float(tot[2 * c])))  # (one host read per BatchNorm: the synchronised mode is not the benchmarked one)
        part = tot[:2 * c].to(torch.float32).view(1, 2, c).contiguous()
        st.m_total = m
    nblk = part.shape[0]
    nb = lib.simhand_bn_finalize_workspace_bytes(nblk, c)
    ws = torch.empty(nb, dtype=torch.uint8, device=part.device)
    # one launch: the block that draws the last ticket finalizes (the ticket word is this stream's, zero between launches)
    check(lib.simhand_bn_finalize_ticket(_ptr(part), nblk, m, c, _ptr(gamma), _ptr(beta), _ptr(pre_bias), eps, momentum,
                                         _ptr(running_mean), _ptr(running_var), _ptr(nbt), _ptr(st.mean), _ptr(st.invstd),
                                         _ptr(st.scale), _ptr(st.shift), _ptr(ws), nb, _ptr(_ticket(part.device)), _stream()), "bn_finalize_ticket")
    return st


def bn_eval_state(c: int, gamma, beta, running_mean, running_var, eps: float = 1e-5) -> BNState:
    lib = _lib_dev()
    st = BNState(c, running_mean.device)
    check(lib.simhand_bn_eval_params(_ptr(gamma), _ptr(beta), _ptr(running_mean), _ptr(running_var), eps, c, _ptr(st.scale),
                                     _ptr(st.shift), _stream()), "bn_eval_params")
    return st


def bn_apply(y, st: BNState, m: int, c: int, relu: bool, residual=None, out=None, want_mask: bool = False):
    """a = act(y*scale + shift (+ residual)).  want_mask: also return the ReLU bit mask ([m][c/VE] uint8) that the
    backward of a residual unit reads instead of the activation."""
    lib = _lib_dev()
    a = torch.empty_like(y) if out is None else out
    mask = None
    if want_mask:
        ve = 4 if y.dtype == torch.float32 else 8
        mask = torch.empty(m, c // ve, dtype=torch.uint8, device=y.device)
    check(lib.simhand_bn_apply(_ptr(y), _ptr(st.scale), _ptr(st.shift), _ptr(residual), int(relu), _ptr(a), _ptr(mask), m, c,
                               dt(y.dtype), _stream()), "bn_apply")
    return (a, mask) if want_mask else a


def bn_backward(da, a, y, st: BNState, gamma, m: int, c: int, relu: bool, want_dres: bool, mask_from_y: bool = False,
                relu_mask=None, raw_partial=None, apply: bool = True, fp8_scaler: "Optional[FP8Scaler]" = None, want_coefs: bool = False):
    """Returns (dy, dres or None, dgamma, dbeta).  raw_partial: (sum g, sum g*y) tiles from conv2d_dgrad_fused -- the
    standalone partial-sum pass is skipped.  mask_from_y: the unit had no residual add, so the ReLU mask is
    recomputed from y (the stored activation is not read).  relu_mask: bit mask from bn_apply(want_mask=True) --
    g = da * bit, whatever `relu` says about this unit's own activation (used for residual units and for the
    downsample branch, whose incoming gradient is the block output's masked gradient)."""
    lib = _lib_dev()
    dev = y.device
    mode = 3 if relu_mask is not None else (0 if not relu else (2 if mask_from_y else 1))
    aa = relu_mask if mode == 3 else (a if mode == 1 else None)
    dg = torch.empty(c, dtype=torch.float32, device=dev)
    db = torch.empty(c, dtype=torch.float32, device=dev)
    coefs = None
    if raw_partial is not None:
        # sums of g and g*y came out of the epilogue of the dgrad that produced `da` (conv2d_dgrad_fused)
        nb = lib.simhand_bn_bwd_finalize_raw_workspace_bytes(raw_partial.shape[0], c)
        ws = torch.empty(nb, dtype=torch.uint8, device=dev)
        if want_coefs and not apply and _BN_SYNC is None and not _FOLD_LEGACY:
            # the consumer applies dy = A g - B y + C itself: (A, B, C) leave the launch that folds the sums (round 6)
            coefs = torch.empty(3, c, dtype=torch.float32, device=dev)
            check(lib.simhand_bn_bwd_finalize_raw_coefs(_ptr(raw_partial), raw_partial.shape[0], c, _ptr(st.mean), _ptr(st.invstd), _ptr(gamma), m,
                                                        _ptr(dg), _ptr(db), _ptr(coefs), _ptr(ws), nb, _stream()), "bn_bwd_finalize_raw_coefs")
        else:
            check(lib.simhand_bn_bwd_finalize_raw(_ptr(raw_partial), raw_partial.shape[0], c, _ptr(st.mean), _ptr(st.invstd), _ptr(dg),
                                                  _ptr(db), _ptr(ws), nb, _stream()), "bn_bwd_finalize_raw")
    else:
        nblk = lib.simhand_bn_stat_blocks(m, c)
        part = torch.empty(nblk, 2, c, dtype=torch.float32, device=dev)
        check(lib.simhand_bn_bwd_partial(_ptr(da), _ptr(aa), _ptr(y), _ptr(st.mean), _ptr(st.invstd), _ptr(st.scale), _ptr(st.shift), mode, m, c,
                                         dt(y.dtype), _ptr(part), _stream()), "bn_bwd_partial")
        check(lib.simhand_bn_bwd_finalize(_ptr(part), nblk, c, _ptr(dg), _ptr(db), _stream()), "bn_bwd_finalize")
    if not apply:  # sums only: the caller fuses the apply into a consumer (conv2d_wgrad_bnbwd)
        if _BN_SYNC is not None:
            raise RuntimeError("bn_backward(apply=False): the fused-apply consumers take local sums; not available under bn_sync")
        if want_coefs:
            if coefs is None:  # (no raw partial sums, or the legacy chain: the coefficient launch of its own)
                return None, None, dg, db, bn_bwd_coefs(st, gamma, dg, db, m)
            return None, None, dg, db, (coefs[0], coefs[1], coefs[2])
        return None, None, dg, db
    dg_l, db_l = dg, db
    if _BN_SYNC is not None:  # dy needs the sums over every rank's positions; the returned parameter gradients stay local
        # (the apply kernels use the sums only as dgamma / m and dbeta / m, with m = the LOCAL row count they also iterate over:
        # they get the global sums scaled by m / m_total)
        both = torch.stack((dg, db))
        _BN_SYNC(both)
        both *= float(m) / float(st.m_total if st.m_total is not None else m)
        dg, db = both[0], both[1]
    if fp8_scaler is not None and fp8_scaler.calls > 0 and not want_dres:
        # the apply pass also emits dy's e4m3 codes (operand of the fp8 data gradient) + its amax; returned as a 5th value
        dy = torch.empty_like(y)
        q = torch.empty(y.shape, dtype=torch.uint8, device=dev)
        check(lib.simhand_bn_bwd_apply_fp8(_ptr(da), _ptr(aa), _ptr(y), _ptr(st.mean), _ptr(st.invstd), _ptr(gamma), _ptr(dg), _ptr(db),
                                           _ptr(st.scale), _ptr(st.shift), mode, _ptr(dy), _ptr(q), _ptr(fp8_scaler.state), _ptr(fp8_scaler.amax_bits),
                                           m, c, _stream()), "bn_bwd_apply_fp8")
        fp8_scaler._update(1)
        fp8_scaler.calls += 1
        return dy, None, dg_l, db_l, q
    dy = torch.empty_like(y)
    dres = torch.empty_like(y) if want_dres else None
    check(lib.simhand_bn_bwd_apply(_ptr(da), _ptr(aa), _ptr(y), _ptr(st.mean), _ptr(st.invstd), _ptr(gamma), _ptr(dg), _ptr(db),
                                   _ptr(st.scale), _ptr(st.shift), mode, _ptr(dy), _ptr(dres), m, c, dt(y.dtype), _stream()), "bn_bwd_apply")
    if fp8_scaler is not None:  # first call of a delayed site: calibrate with the two-pass form
        return dy, dres, dg_l, db_l, fp8_scaler.quantize(dy)
    return dy, dres, dg_l, db_l


def bn_relu_maxpool_fwd(y: torch.Tensor, st: BNState, want_winner: bool = False):
    """Stem: MaxPool(3,2,1)(ReLU(BN(y))) in one pass; y is [n][h][w][c].  Returns (pooled, winner index) and, with
    want_winner, the raw y of each window's winning tap (for maxpool_bn_backward's pooled-size statistics pass)."""
    lib = _lib_dev()
    n, h, w, c = y.shape
    ho, wo = (h - 1) // 2 + 1, (w - 1) // 2 + 1
    out = torch.empty(n, ho, wo, c, dtype=y.dtype, device=y.device)
    idx = torch.empty(n, ho, wo, c, dtype=torch.uint8, device=y.device)
    ywin = torch.empty_like(out) if want_winner else None
    check(lib.simhand_bn_relu_maxpool_fwd(_ptr(y), _ptr(st.scale), _ptr(st.shift), _ptr(out), _ptr(idx), _ptr(ywin), n, h, w, c,
                                          dt(y.dtype), _stream()), "bn_relu_maxpool_fwd")
    return (out, idx, ywin) if want_winner else (out, idx)


def maxpool_bn_backward(dz: torch.Tensor, idx: torch.Tensor, y: torch.Tensor, st: BNState, gamma, ywin: Optional[torch.Tensor] = None):
    """Backward of bn_relu_maxpool_fwd: (dy, dgamma, dbeta); dz is the gradient of the pooled output.  ywin (the raw y of
    the winning taps, from bn_relu_maxpool_fwd(want_winner=True)): the BatchNorm-backward sums are taken over the pooled
    tensors (each pooled gradient lands on exactly its winner) instead of a gather pass over the 4x larger y."""
    lib = _lib_dev()
    n, h, w, c = y.shape
    m = n * h * w
    dev = y.device
    if ywin is not None:
        mo = dz.numel() // c
        nblk = lib.simhand_bn_stat_blocks(mo, c)
        part = torch.empty(nblk, 2, c, dtype=torch.float32, device=dev)
        check(lib.simhand_bn_bwd_partial(_ptr(dz), None, _ptr(ywin), _ptr(st.mean), _ptr(st.invstd), _ptr(st.scale), _ptr(st.shift), 2, mo, c,
                                         dt(y.dtype), _ptr(part), _stream()), "bn_bwd_partial (pooled)")
    else:
        nblk = lib.simhand_bn_stat_blocks(m, c)
        part = torch.empty(nblk, 2, c, dtype=torch.float32, device=dev)
        check(lib.simhand_maxpool_bn_bwd_partial(_ptr(dz), _ptr(idx), _ptr(y), _ptr(st.mean), _ptr(st.invstd), _ptr(st.scale), _ptr(st.shift),
                                                 n, h, w, c, dt(y.dtype), _ptr(part), _stream()), "maxpool_bn_bwd_partial")
    dg = torch.empty(c, dtype=torch.float32, device=dev)
    db = torch.empty(c, dtype=torch.float32, device=dev)
    check(lib.simhand_bn_bwd_finalize(_ptr(part), nblk, c, _ptr(dg), _ptr(db), _stream()), "bn_bwd_finalize")
    dg_l, db_l = dg, db
    if _BN_SYNC is not None:
        # the apply kernel divides the two sums by its LOCAL position count m: hand it the global sums scaled by m / m_total
        both = torch.stack((dg, db))
        _BN_SYNC(both)
        both *= float(m) / float(st.m_total if st.m_total is not None else m)
        dg, db = both[0], both[1]
    dy = torch.empty_like(y)
    check(lib.simhand_maxpool_bn_bwd_apply(_ptr(dz), _ptr(idx), _ptr(y), _ptr(st.mean), _ptr(st.invstd), _ptr(gamma), _ptr(dg), _ptr(db),
                                           _ptr(st.scale), _ptr(st.shift), _ptr(dy), n, h, w, c, dt(y.dtype), _stream()), "maxpool_bn_bwd_apply")
    return dy, dg_l, db_l


# ------------------------------------------------------------------------ pools
def maxpool_fwd(x: torch.Tensor):
    lib = _lib_dev()
    n, h, w, c = x.shape
    ho, wo = (h - 1) // 2 + 1, (w - 1) // 2 + 1
    y = torch.empty(n, ho, wo, c, dtype=x.dtype, device=x.device)
    idx = torch.empty(n, ho, wo, c, dtype=torch.uint8, device=x.device)
    check(lib.simhand_maxpool3x3s2_fwd(_ptr(x), _ptr(y), _ptr(idx), n, h, w, c, dt(x.dtype), _stream()), "maxpool_fwd")
    return y, idx


def maxpool_bwd(dy: torch.Tensor, idx: torch.Tensor, in_shape) -> torch.Tensor:
    lib = _lib_dev()
    n, h, w, c = in_shape
    dx = torch.empty(n, h, w, c, dtype=dy.dtype, device=dy.device)
    check(lib.simhand_maxpool3x3s2_bwd(_ptr(dy), _ptr(idx), _ptr(dx), n, h, w, c, dt(dy.dtype), _stream()), "maxpool_bwd")
    return dx


def subsample2(x: torch.Tensor) -> torch.Tensor:
    """x[:, ::2, ::2, :] as a dense tensor (the pixels a stride-2 1x1 convolution reads)."""
    lib = _lib_dev()
    n, h, w, c = x.shape
    y = torch.empty(n, (h + 1) // 2, (w + 1) // 2, c, dtype=x.dtype, device=x.device)
    check(lib.simhand_subsample2(_ptr(x), _ptr(y), n, h, w, c, dt(x.dtype), _stream()), "subsample2")
    return y


def scatter2_add(src: torch.Tensor, dx: torch.Tensor, mask=None) -> torch.Tensor:
    """dx[:, ::2, ::2, :] = gate(dx[:, ::2, ::2, :] + src) in place (gate: optional ReLU bit mask over dx's pixels)."""
    lib = _lib_dev()
    n, h, w, c = dx.shape
    check(lib.simhand_scatter2_add(_ptr(src), _ptr(dx), _ptr(mask), n, h, w, c, dt(dx.dtype), _stream()), "scatter2_add")
    return dx


def avgpool_fwd(x: torch.Tensor) -> torch.Tensor:
    lib = _lib_dev()
    n, h, w, c = x.shape
    y = torch.empty(n, c, dtype=x.dtype, device=x.device)
    check(lib.simhand_avgpool_fwd(_ptr(x), _ptr(y), n, h * w, c, dt(x.dtype), _stream()), "avgpool_fwd")
    return y


def avgpool_bwd(dy: torch.Tensor, in_shape, mask=None) -> torch.Tensor:
    """mask: ReLU bit mask of the pooled tensor -- the gradient leaves already gated (== apply_relu_bitmask(avgpool_bwd(dy), mask))."""
    lib = _lib_dev()
    n, h, w, c = in_shape
    dx = torch.empty(n, h, w, c, dtype=dy.dtype, device=dy.device)
    check(lib.simhand_avgpool_bwd_masked(_ptr(dy), _ptr(mask), _ptr(dx), n, h * w, c, dt(dy.dtype), _stream()), "avgpool_bwd")
    return dx


def bn_fold_fwd(w: torch.Tensor, round_bf16: bool, s2, t2, m: int, gamma, beta, running_mean, running_var, nbt, eps: float = 1e-5,
                momentum: float = 0.1):
    """Batch statistics of y = a W^T from the Gram matrix of a (see simhand_bn_fold_fwd).  w: fp32 [cc][cw].
    Returns (BNState, ws2 = W s2)."""
    lib = _lib_dev()
    cc, cw = w.shape
    st = BNState(cc, w.device)
    ws2 = torch.empty(cc, cw, dtype=torch.float32, device=w.device)
    nb = lib.simhand_bn_fold_workspace_bytes(cc, cw)
    wsp = torch.empty(nb, dtype=torch.uint8, device=w.device)
    check(lib.simhand_bn_fold_fwd(_ptr(w), int(round_bf16), _ptr(s2), _ptr(t2), cc, cw, m, _ptr(gamma), _ptr(beta), eps, momentum,
                                  _ptr(running_mean), _ptr(running_var), _ptr(nbt), _ptr(st.mean), _ptr(st.invstd), _ptr(st.scale),
                                  _ptr(st.shift), _ptr(ws2), _ptr(wsp), nb, _stream()), "bn_fold_fwd")
    return st, ws2


def bn_fold_bwd(w: torch.Tensor, round_bf16: bool, gmat, s, ws2, t2, st: BNState, gamma, m: int, dtype: torch.dtype):
    """Parameter-sized part of the folded BatchNorm backward (see simhand_bn_fold_bwd).
    Returns (dgamma, dbeta, dw [cc][cw], wa CRSK [cw][cc], wm CRSK [cw][cw], bias [cw])."""
    lib = _lib_dev()
    cc, cw = w.shape
    dev = w.device
    f32 = torch.float32
    dg = torch.empty(cc, dtype=f32, device=dev)
    db = torch.empty(cc, dtype=f32, device=dev)
    dw = torch.empty(cc, cw, dtype=f32, device=dev)
    wa = torch.empty(cw, cc, dtype=dtype, device=dev)
    wm = torch.empty(cw, cw, dtype=dtype, device=dev)
    bw = torch.empty(cc, cw, dtype=f32, device=dev)
    cco = torch.empty(cc, dtype=f32, device=dev)
    bias = torch.empty(cw, dtype=f32, device=dev)
    nb = lib.simhand_bn_fold_workspace_bytes(cc, cw)
    wsp = torch.empty(nb, dtype=torch.uint8, device=dev)
    check(lib.simhand_bn_fold_bwd(_ptr(w), int(round_bf16), _ptr(gmat), _ptr(s), _ptr(ws2), _ptr(t2), _ptr(st.mean), _ptr(st.invstd),
                                  _ptr(gamma), cc, cw, m, _ptr(dg), _ptr(db), _ptr(dw), _ptr(wa), _ptr(bw), _ptr(cco), _ptr(wm), _ptr(bias),
                                  dt(dtype), _ptr(wsp), nb, _stream()), "bn_fold_bwd")
    return dg, db, dw, wa, wm, bias


def apply_relu_bitmask(x: torch.Tensor, mask: torch.Tensor) -> torch.Tensor:
    """x gated by a ReLU bit mask ([rows][c / VE] bytes from bn_apply / conv2d_fwd_bnact)."""
    lib = _lib_dev()
    c = x.shape[-1]
    m = x.numel() // c
    out = torch.empty_like(x)
    check(lib.simhand_apply_relu_bitmask(_ptr(x), _ptr(mask), _ptr(out), m, c, dt(x.dtype), _stream()), "apply_relu_bitmask")
    return out


_UNIT_COEF: dict = {}


def bn_channel_sums(raw_partial: torch.Tensor, c: int) -> torch.Tensor:
    """Column 0 of the tile partials of a fused dgrad epilogue (sum over tiles, fp64 fold): [c] fp32."""
    lib = _lib_dev()
    dev = raw_partial.device
    out = torch.empty(2, c, dtype=torch.float32, device=dev)
    key = (c, dev)
    if key not in _UNIT_COEF:  # constant (mean = 0, invstd = 1) vectors: built once per channel count, not per call
        _UNIT_COEF[key] = (torch.zeros(c, dtype=torch.float32, device=dev), torch.ones(c, dtype=torch.float32, device=dev))
    zero, one = _UNIT_COEF[key]
    nb = lib.simhand_bn_bwd_finalize_raw_workspace_bytes(raw_partial.shape[0], c)
    ws = torch.empty(nb, dtype=torch.uint8, device=dev)
    # finalize_raw with mean = 0, invstd = 1: "dbeta" = sum of column 0, "dgamma" = sum of column 1
    check(lib.simhand_bn_bwd_finalize_raw(_ptr(raw_partial), raw_partial.shape[0], c, _ptr(zero), _ptr(one), _ptr(out[1]), _ptr(out[0]),
                                          _ptr(ws), nb, _stream()), "bn_bwd_finalize_raw")
    return out[0]


def colsum(x: torch.Tensor, m: int, c: int) -> torch.Tensor:
    lib = _lib_dev()
    nblk = lib.simhand_bn_stat_blocks(m, c)
    part = torch.empty((2 * nblk + 1) * c, dtype=torch.float32, device=x.device)
    out = torch.empty(c, dtype=torch.float32, device=x.device)
    check(lib.simhand_colsum(_ptr(x), m, c, dt(x.dtype), _ptr(part), _ptr(out), _stream()), "colsum")
    return out


# -------------------------------------------------------------------------- fp8
FP8_HISTORY = 16


class FP8Scaler:
    """Per-tensor scale of one quantisation site (a weight tensor or an activation edge) -- the mixed-precision policy of
    the fp8 slice (the reference's fp16 GradScaler, src/experiments/main.py:158-159, has no fp8 analogue):

    * e4m3 for forward operands, value range +-448; q = e4m3(clamp(v * scale));
    * weights: CURRENT scaling -- amax of the fp32 master at pack time (weights change once per optimizer step);
    * activations: DELAYED scaling -- the scale used now comes from the amax ring (last FP8_HISTORY calls, max), the tensor's
      own amax is recorded by the same quantize pass for the next call: one pass over the tensor, no host sync.  The first
      call has no history and falls back to a current-scaling pre-pass.  margin = 1 bit of headroom (scale / 2);
    * everything the scale touches stays on the device (state vector), nothing is read back per step."""

    def __init__(self, device, delayed: bool, margin_bits: int = 1):
        lib = _lib_dev()
        self.delayed, self.margin = delayed, float(2 ** margin_bits)
        self.state = torch.zeros(lib.simhand_fp8_state_floats(FP8_HISTORY), dtype=torch.float32, device=device)
        self.state[0] = 1.0
        self.state[1] = 1.0
        self.amax_bits = torch.zeros(1, dtype=torch.int32, device=device)
        self.calls = 0

    def _update(self, mode: int):
        """mode 0: current scaling (before the quantisation it serves); 1: a delayed site's update BEHIND its quantisation (state[1] keeps the
        reciprocal of the scale the codes were made with); 2: a delayed site's first update, before its first quantisation."""
        check(_lib_dev().simhand_fp8_scale_update(_ptr(self.state), _ptr(self.amax_bits), FP8_HISTORY, self.margin, int(mode), _stream()),
              "fp8_scale_update")

    def quantize(self, x: torch.Tensor) -> torch.Tensor:
        """x (bf16 / fp32, contiguous, numel % 16 == 0) -> uint8 tensor of e4m3 codes with this site's scale."""
        lib = _lib_dev()
        q = torch.empty(x.shape, dtype=torch.uint8, device=x.device)
        if not self.delayed or self.calls == 0:
            check(lib.simhand_fp8_amax(_ptr(x), x.numel(), dt(x.dtype), _ptr(self.amax_bits), _stream()), "fp8_amax")
            self._update(2 if self.delayed else 0)  # current scaling (and, for a delayed site, the ring's first entry)
            check(lib.simhand_fp8_quantize(_ptr(x), _ptr(q), x.numel(), dt(x.dtype), _ptr(self.state), None, _stream()), "fp8_quantize")
        else:
            check(lib.simhand_fp8_quantize(_ptr(x), _ptr(q), x.numel(), dt(x.dtype), _ptr(self.state), _ptr(self.amax_bits), _stream()),
                  "fp8_quantize")
            self._update(1)  # this call's amax enters the ring: the scale of the NEXT call
        self.calls += 1
        return q

    def bn_apply_quantize(self, y: torch.Tensor, st: "BNState", relu: bool):
        """(a, q): a = act(y*scale + shift) in bf16 and its e4m3 codes with this site's scale from ONE pass over y (simhand_bn_apply_fp8);
        bit-identical to bn_apply followed by quantize.  The first call of a delayed site has no history yet and runs the two-pass form."""
        c = y.shape[-1]
        m = y.numel() // c
        if not self.delayed or self.calls == 0:
            a = bn_apply(y, st, m, c, relu)
            return a, self.quantize(a)
        lib = _lib_dev()
        a = torch.empty_like(y)
        q = torch.empty(y.shape, dtype=torch.uint8, device=y.device)
        check(lib.simhand_bn_apply_fp8(_ptr(y, torch.bfloat16), _ptr(st.scale), _ptr(st.shift), int(relu), _ptr(a), _ptr(q), _ptr(self.state),
                                       _ptr(self.amax_bits), m, c, _stream()), "bn_apply_fp8")
        self._update(1)
        self.calls += 1
        return a, q

    def pack_weights(self, w_oihw: torch.Tensor) -> torch.Tensor:
        lib = _lib_dev()
        k, c, r, s = w_oihw.shape
        wq = torch.empty(k, r * s * c, dtype=torch.uint8, device=w_oihw.device)
        w = w_oihw.detach().contiguous()
        check(lib.simhand_fp8_amax(_ptr(w, _F32), w.numel(), _lib.SH_F32, _ptr(self.amax_bits), _stream()), "fp8_amax")
        self._update(0)
        check(lib.simhand_fp8_pack_krsc(_ptr(w), _ptr(wq), k, c, r, s, _ptr(self.state), _stream()), "fp8_pack_krsc")
        return wq


def fp8_pack_crsk(scaler: FP8Scaler, w_oihw: torch.Tensor) -> torch.Tensor:
    """e4m3 CRSK weights [cin][r][s][cout] for the fp8 data gradient: the KRSC packer applied to the (cin, cout)-transposed filter."""
    return scaler.pack_weights(w_oihw.detach().permute(1, 0, 2, 3).contiguous())


def conv2d_wgrad_fp8_pays(d: ConvDesc) -> bool:
    """The layers whose weight gradient runs on e4m3 operands in the fp8 configuration (3x3 / stride 1, >= 256 channels on both sides)."""
    return bool(_lib.load().simhand_conv2d_wgrad_fp8_pays(C.byref(d)))


def conv2d_wgrad_fp8(d: ConvDesc, x_q, dy_q, x_state: torch.Tensor, dy_state: torch.Tensor) -> torch.Tensor:
    """fp32 weight.grad [cout][cin][3][3] from the e4m3 codes of the activation and of dy.  x_state / dy_state: fp32 device vectors whose
    element [1] is 1 / (the scale THESE codes were made with) -- an FP8Scaler's .state right behind the quantisation, or a snapshot of its
    first two floats taken then (what the engine keeps for the forward's codes: the live state moves on with every later quantisation)."""
    lib = _lib_dev()
    nb = lib.simhand_conv2d_wgrad_fp8_workspace_bytes(C.byref(d))
    ws = torch.empty(nb, dtype=torch.uint8, device=x_q.device)
    dw = torch.empty(d.cout, d.cin, 3, 3, dtype=torch.float32, device=x_q.device)
    check(lib.simhand_conv2d_wgrad_fp8(C.byref(d), _ptr(x_q, torch.uint8), _ptr(dy_q, torch.uint8), _ptr(x_state, _F32), _ptr(dy_state, _F32),
                                       _ptr(dw), _ptr(ws), nb, _stream()), "conv2d_wgrad_fp8")
    return dw


def conv2d_fwd_fp8_supported(d: ConvDesc) -> bool:
    return bool(_lib.load().simhand_conv2d_fwd_fp8_supported(C.byref(d)))


def conv2d_dgrad_fp8_pays(d: ConvDesc) -> bool:
    """The layers whose data gradient runs on the e4m3 variant of the 256 x 256 kernel in the fp8 configuration (3x3, >= 256 channels)."""
    return bool(_lib.load().simhand_conv2d_dgrad_fp8_pays(C.byref(d)))


def conv2d_fwd_fp8_pays(d: ConvDesc) -> bool:
    """The layers whose fp8 forward measured faster than their bf16 kernel (the engine's default fp8 set)."""
    return bool(_lib.load().simhand_conv2d_fwd_fp8_pays(C.byref(d)))


def conv2d_fwd_fp8(d: ConvDesc, x_q, w_q, x_scaler: FP8Scaler, w_scaler: FP8Scaler, want_stats: bool = True):
    """y (bf16) = conv(x_q, w_q) / (scale_x * scale_w) on the e4m3 scaled-MFMA kernel; BN partial sums as conv2d_fwd."""
    lib = _lib_dev()
    y = torch.empty(d.n, d.ho, d.wo, d.cout, dtype=torch.bfloat16, device=x_q.device)
    part = torch.empty(lib.simhand_conv2d_fwd_fp8_stat_blocks(C.byref(d)), 2, d.cout, dtype=torch.float32, device=x_q.device) if want_stats else None
    check(lib.simhand_conv2d_fwd_fp8(C.byref(d), _ptr(x_q, torch.uint8), _ptr(w_q, torch.uint8), _ptr(x_scaler.state), _ptr(w_scaler.state), _ptr(y),
                                     _ptr(part), _stream()), "conv2d_fwd_fp8")
    return y, part


# ---------------------------------------------------------------- batch producer
def augment_batch(images_u8: torch.Tensor, joints: torch.Tensor, angle, crop_margin, jitter, hsab, out_hw=(128, 128), extra: Optional[dict] = None):
    """images [n][h][w][3] uint8, joints [n][21][3] fp32, angle [n] fp32 or None, crop_margin [n] fp32, jitter [n][2] int32,
    hsab [n][4] fp32 or None -> (images [n][3][oh][ow] fp32 normalised, joints_aug [n][21][3], rec [n][6] int32).
    extra (the coin-flip operations, simhand_augment_batch_ex): dict with flags [n] int32 (bit 0 sobel, 1 cut-out, 2 blur, 3 noise,
    4 colour drop) and, as used, cut_box [n][4] int32, cut_fill [n] uint8, blur_sigma [n] fp32, blur_k (kx, ky), noise [n][oh][ow][3]
    fp32 standard-normal draws, noise_std; any_* (host booleans: does any sample carry the bit)."""
    lib = _lib_dev()
    n, h, w, _ = images_u8.shape
    oh, ow = out_hw
    dev = images_u8.device
    out = torch.empty(n, 3, oh, ow, dtype=torch.float32, device=dev)
    ja = torch.empty(n, 21, 3, dtype=torch.float32, device=dev)
    rec = torch.empty(n, 6, dtype=torch.int32, device=dev)
    if extra is not None:
        ex = _lib.AugmentExtra()
        ex.flags = _ptr(extra["flags"], torch.int32).value
        ex.cut_box = _ptr(extra.get("cut_box"), torch.int32).value
        ex.cut_fill = _ptr(extra.get("cut_fill"), torch.uint8).value
        ex.blur_sigma = _ptr(extra.get("blur_sigma"), _F32).value
        ex.noise = _ptr(extra.get("noise"), _F32).value
        ex.noise_std = float(extra.get("noise_std", 0.0))
        ex.blur_kx, ex.blur_ky = (int(v) for v in extra.get("blur_k", (1, 1)))
        ex.any_sobel, ex.any_cut_out = int(bool(extra.get("any_sobel"))), int(bool(extra.get("any_cut_out")))
        ex.any_blur, ex.any_noise = int(bool(extra.get("any_blur"))), int(bool(extra.get("any_noise")))
        pre = ex.any_sobel or ex.any_cut_out or ex.any_blur
        nb = lib.simhand_augment_workspace_bytes_ex(n, h, w, int(pre), ex.any_blur)
        ws = torch.empty(nb, dtype=torch.uint8, device=dev)
        check(lib.simhand_augment_batch_ex(_ptr(images_u8, torch.uint8), _ptr(joints, _F32), _ptr(angle, _F32), _ptr(crop_margin, _F32),
                                           _ptr(jitter, torch.int32), _ptr(hsab, _F32), C.byref(ex), n, h, w, ow, oh, _ptr(out), _ptr(ja), _ptr(rec),
                                           _ptr(ws), nb, _stream()), "augment_batch_ex")
        return out, ja, rec
    nb = lib.simhand_augment_workspace_bytes(n)
    ws = torch.empty(nb, dtype=torch.uint8, device=dev)
    check(lib.simhand_augment_batch(_ptr(images_u8, torch.uint8), _ptr(joints, _F32), _ptr(angle, _F32), _ptr(crop_margin, _F32),
                                    _ptr(jitter, torch.int32), _ptr(hsab, _F32), n, h, w, ow, oh, _ptr(out), _ptr(ja), _ptr(rec), _ptr(ws), nb,
                                    _stream()), "augment_batch")
    return out, ja, rec


# -------------------------------------------------------------------- optimizer
def lars_adam_step(param, grad, exp_avg, exp_avg_sq, step: int, lr: float, weight_decay: float, use_lars: bool,
                   betas=(0.9, 0.999), adam_eps: float = 1e-8, lars_eta: float = 0.02, lars_eps: float = 1e-8,
                   lars_clip: bool = True) -> None:
    lib = _lib_dev()
    n = param.numel()
    nblk = max(1, min(256, (n + 4095) // 4096))
    pp = gp = None
    if use_lars:
        pp = torch.empty(nblk, dtype=torch.float32, device=param.device)
        gp = torch.empty(nblk, dtype=torch.float32, device=param.device)
        check(lib.simhand_sumsq_partial(_ptr(param), n, _ptr(pp), nblk, _stream()), "sumsq_partial")
        check(lib.simhand_sumsq_partial(_ptr(grad), n, _ptr(gp), nblk, _stream()), "sumsq_partial")
    check(lib.simhand_lars_adam_step(_ptr(param), _ptr(grad), _ptr(exp_avg), _ptr(exp_avg_sq), n, _ptr(pp), _ptr(gp), nblk, lr,
                                     betas[0], betas[1], adam_eps, weight_decay, lars_eta, lars_eps, int(lars_clip), int(use_lars),
                                     step, _stream()), "lars_adam_step")


OPT_TENSOR_DTYPE = np.dtype([("param", "<u8"), ("grad", "<u8"), ("exp_avg", "<u8"), ("exp_avg_sq", "<u8"), ("count", "<i8"),
                             ("first_chunk", "<i4"), ("n_chunks", "<i4"), ("lr", "<f4"), ("weight_decay", "<f4"), ("bc1", "<f4"),
                             ("bc2_sqrt", "<f4"), ("use_lars", "<i4"), ("reserved", "<i4")])  # == sh_opt_tensor


class LarsAdamPlan:
    """Chunk table (device, built once per parameter list) for ``lars_adam_multi``."""

    def __init__(self, counts, device):
        lib = _lib_dev()
        ch = lib.simhand_opt_chunk_elems()
        self.counts = [int(c) for c in counts]
        self.first, self.nchunks, pairs = [], [], []
        for t, c in enumerate(self.counts):
            k = max(1, (c + ch - 1) // ch)
            self.first.append(len(pairs))
            self.nchunks.append(k)
            pairs.extend((t, j) for j in range(k))
        self.n_chunks = len(pairs)
        self.total = sum(self.counts)
        self.chunks = torch.tensor(pairs, dtype=torch.int32).reshape(-1, 2).contiguous().to(device)
        self.partials = torch.empty(2 * self.n_chunks, dtype=torch.float32, device=device)


def lars_adam_multi(plan: LarsAdamPlan, records: "np.ndarray", betas=(0.9, 0.999), adam_eps: float = 1e-8, lars_eta: float = 0.02,
                    lars_eps: float = 1e-8, lars_clip: bool = True, found_inf: Optional[torch.Tensor] = None) -> None:
    """records: OPT_TENSOR_DTYPE array, one per tensor of the plan (device pointers as integers).  found_inf: GradScaler's device
    flag (fp32 [1]); non-zero -> the update launch leaves parameters and moments untouched (simhand_lars_adam_multi_guarded)."""
    lib = _lib_dev()
    assert records.dtype == OPT_TENSOR_DTYPE and len(records) == len(plan.counts)
    host = torch.from_numpy(records.view(np.uint8)).pin_memory()  # caching host allocator: safe to reuse across async copies
    table = host.to(plan.chunks.device, non_blocking=True)
    if found_inf is not None:
        assert found_inf.dtype == torch.float32 and found_inf.numel() == 1 and found_inf.device == plan.chunks.device
    check(lib.simhand_lars_adam_multi_guarded(_ptr(table), len(records), _ptr(plan.chunks), plan.n_chunks, _ptr(plan.partials), betas[0],
                                              betas[1], adam_eps, lars_eta, lars_eps, int(lars_clip), plan.total, _ptr(found_inf), _stream()),
          "lars_adam_multi")


# ----------------------------------------------------------------- route counters
def route_reset() -> None:
    _lib.load().simhand_route_reset()


def route_counts() -> dict:
    """{route name: launches since the last reset} -- which hand-written kernel each call was dispatched to."""
    buf = (C.c_int64 * _lib.ROUTE_COUNT)()
    check(_lib.load().simhand_route_counts(buf), "route_counts")
    return {name: int(buf[i]) for i, name in enumerate(_lib.ROUTES)}


# include/simhand_hip.h enum sh_test_switch (the kernel-selection switches that were SIMHAND_* environment variables in round 3)
TEST_SWITCHES = {"BN_GRID_APPLY": 0, "BN_GRID_BWD": 1, "R128": 2, "G1_PF": 3, "G1_CHAIN": 4, "G1_LT": 5, "FUSE_S2": 6, "WG_DMA": 7,
                 "WG3_S2": 8, "WG_BIG": 9, "WG_WIDE": 10, "STEM_WG256": 11, "STEM_RING": 12, "STEM_RING_LT": 13, "STEM_WG_RING": 14, "N128": 15, "FOLD_LEGACY": 16}


_FOLD_LEGACY = False  # mirror of SH_SW_FOLD_LEGACY for the wrappers below whose merged form is another ENTRY POINT, not another kernel


def test_switch(name: str, value: int) -> None:
    """simhand_test_switch: value < 0 = the built-in default.  Tests / A-B timing only (bench.py --switch NAME=V)."""
    global _FOLD_LEGACY
    check(_lib.load().simhand_test_switch(TEST_SWITCHES[name.upper()], int(value)), "test_switch")
    if name.upper() == "FOLD_LEGACY":
        _FOLD_LEGACY = int(value) > 0


def hooks_reset() -> None:
    """Every test / tuning hook of the library back to its default."""
    global _FOLD_LEGACY
    _lib.load().simhand_test_hooks_reset()
    _FOLD_LEGACY = False


# --------------------------------------------------------------------- profiler
def prof_enable(on: bool) -> None:
    _lib.load().simhand_prof_enable(int(on))


def prof_set_classes(names=None) -> None:
    """Restrict the event profiler to the named kernel classes (None = all)."""
    mask = 0xFFFFFFFF if names is None else sum(1 << _lib.PROF_CLASSES.index(n) for n in names)
    _lib.load().simhand_prof_set_classes(mask)


def prof_reset() -> None:
    _lib.load().simhand_prof_reset()


def prof_records(max_records: int = 4096) -> list:
    """[(class name, ms, algorithmic FLOPs, algorithmic bytes)] of every launch recorded since the last collect / reset, in issue order."""
    lib = _lib.load()
    cls = (C.c_int * max_records)()
    ms = (C.c_double * max_records)()
    fl = (C.c_double * max_records)()
    by = (C.c_double * max_records)()
    n = C.c_int()
    check(lib.simhand_prof_records(max_records, cls, ms, fl, by, C.byref(n)), "prof_records")
    return [(_lib.PROF_CLASSES[cls[i]], ms[i], fl[i], by[i]) for i in range(n.value)]


def prof_collect() -> dict:
    lib = _lib.load()
    n = len(_lib.PROF_CLASSES)
    ms, fl, by = (C.c_double * n)(), (C.c_double * n)(), (C.c_double * n)()
    cnt = (C.c_int64 * n)()
    check(lib.simhand_prof_collect(ms, fl, by, cnt), "prof_collect")
    return {name: {"ms": ms[i], "flops": fl[i], "bytes": by[i], "count": cnt[i]} for i, name in enumerate(_lib.PROF_CLASSES)}
