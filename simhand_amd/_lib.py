"""ctypes binding of ``libsimhand_hip.so`` (the C ABI declared in
``include/simhand_hip.h``).

There is NO CPU fallback: if the shared library is missing, or a compute entry
point is called without a gfx950 device, this module raises.  torch is used by
the callers only as the owner of device memory and streams.
"""
from __future__ import annotations

import ctypes as C
import os
import threading

# torch ships its own HIP runtime (torch/lib/libamdhip64.so); it must be the first one mapped into the
# process, otherwise torch.cuda later reports "No HIP GPUs are available".  Importing torch before the
# CDLL below guarantees that order; libsimhand_hip.so then binds to the already-loaded runtime by soname.
import torch  # noqa: F401

_HERE = os.path.dirname(os.path.abspath(__file__))
# (another build of the same library for an A/B timing run: set_library_paths below)
LIB_PATH = os.path.join(_HERE, "libsimhand_hip.so")
# the fp16-storage build of the same sources (csrc/Makefile, -DSH_H16_FP16): the reference's precision=16
LIB_PATH_F16 = os.path.join(_HERE, "libsimhand_hip_f16.so")


def set_library_paths(bf16: str = None, f16: str = None) -> None:
    """Another build of the library (scripts/build_variant.sh) for a same-box A/B run: call BEFORE the first load() of that format
    (bench.py --lib / --lib-f16).  The package reads no environment variable for this (round 6; SIMHAND_LIB until round 5)."""
    global LIB_PATH, LIB_PATH_F16
    if bf16 is not None:
        if "bf16" in _libs:
            raise SimhandHipError("set_library_paths: the bf16 library is already loaded")
        LIB_PATH = bf16
    if f16 is not None:
        if "f16" in _libs:
            raise SimhandHipError("set_library_paths: the fp16 library is already loaded")
        LIB_PATH_F16 = f16


# enums of include/simhand_hip.h
SH_F32, SH_BF16, SH_FP8_E4M3 = 0, 1, 2
DIST_MODES = {"mpjpe": 0, "w_abs": 1, "w_o_abs": 2, "l2": 3}
WEIGHT_TYPES = {None: 0, "none": 0, "linear": 1, "non_linear": 2, "explicit": 3}
PP_NORM_IN, PP_NORM_OUT, PP_ANGLE_AS_GIVEN = 1, 2, 4
PP_FUSED = 3
PROF_CLASSES = ("conv_fwd", "conv_dgrad", "conv_wgrad", "bn", "pool", "loss", "misc", "optimizer")
# enum sh_route: kernel the library dispatched a call to (index = counter slot)
ROUTES = ("igemm128_fwd", "igemm128_dgrad", "igemm256_fwd", "igemm256_dgrad", "igemm256_tail", "gemm1x1_fwd", "gemm1x1_fwd_bnact",
          "gemm1x1_dgrad", "c64_fwd", "c64_dgrad", "stem_fwd", "fwd_bnact", "dgrad_concat", "dgrad_fused_sums", "dgrad_parity",
          "wgrad3x3", "wgrad_plain", "wgrad_generic", "wgrad_stem", "wgrad_colsum", "bn_fold_fwd", "bn_fold_bwd", "bn_apply",
          "bn_bwd_apply", "stem_bn_pool", "ntxent_fwd", "ntxent_bwd", "fp8_fwd", "fp8_dgrad", "bn_apply_gram", "wgrad_bnbwd", "ntxent_fused_dist",
          "dgrad_dysrc", "fwd_chain", "r128_fwd", "r128_dgrad", "fwd_bnin", "n128_fwd", "n128_dgrad", "fp8_wgrad", "stem_ring_fwd", "stem_ring_wgrad")
ROUTE_COUNT = 42


class SimhandHipError(RuntimeError):
    pass


class AugmentExtra(C.Structure):  # == sh_augment_extra
    _fields_ = [("flags", C.c_void_p), ("cut_box", C.c_void_p), ("cut_fill", C.c_void_p), ("blur_sigma", C.c_void_p), ("noise", C.c_void_p),
                ("noise_std", C.c_float), ("blur_kx", C.c_int), ("blur_ky", C.c_int), ("any_sobel", C.c_int), ("any_cut_out", C.c_int),
                ("any_blur", C.c_int), ("any_noise", C.c_int)]


class NtxentParams(C.Structure):
    _fields_ = [
        ("B", C.c_int), ("dim", C.c_int), ("b_loc", C.c_int), ("pair_off", C.c_int),
        ("weight_type", C.c_int), ("use_wpos", C.c_int), ("use_wneg", C.c_int),
        ("temperature", C.c_float), ("lambda_pos", C.c_float), ("lambda_neg", C.c_float),
    ]


class BnBwdFuse(C.Structure):
    """sh_bn_bwd_fuse (include/simhand_hip.h)."""
    _fields_ = [("y", C.c_void_p), ("scale", C.c_void_p), ("shift", C.c_void_p), ("mask", C.c_void_p),
                ("relu_mode", C.c_int32), ("partial", C.c_void_p)]


class DySrc(C.Structure):
    """sh_dy_src (include/simhand_hip.h)."""
    _fields_ = [("da", C.c_void_p), ("y", C.c_void_p), ("scale", C.c_void_p), ("shift", C.c_void_p), ("coef_a", C.c_void_p),
                ("coef_b", C.c_void_p), ("coef_c", C.c_void_p), ("relu", C.c_int32), ("dy_out", C.c_void_p)]


class DgradOpts(C.Structure):
    """sh_dgrad_opts (include/simhand_hip.h)."""
    _fields_ = [("accumulate", C.c_int32), ("res_grad", C.c_void_p), ("res_mask", C.c_void_p), ("bias", C.c_void_p),
                ("fuse", C.POINTER(BnBwdFuse)), ("x2", C.c_void_p), ("wt2", C.c_void_p), ("c2", C.c_int32),
                ("dy_src", C.POINTER(DySrc)), ("dy_q", C.c_void_p), ("wt_q", C.c_void_p), ("dy_state", C.c_void_p), ("w_state", C.c_void_p),
                ("sub_grad", C.c_void_p)]


class ConvDesc(C.Structure):
    _fields_ = [
        ("n", C.c_int), ("h", C.c_int), ("w", C.c_int), ("cin", C.c_int),
        ("cout", C.c_int), ("r", C.c_int), ("s", C.c_int),
        ("stride", C.c_int), ("pad", C.c_int), ("ho", C.c_int), ("wo", C.c_int), ("dtype", C.c_int),
    ]


_P = C.c_void_p
_I = C.c_int
_L = C.c_int64
_F = C.c_float
_S = C.c_size_t

ABI_VERSION = 5  # include/simhand_hip.h SH_ABI_VERSION this table was written against (checked in load())

# name -> (restype, argtypes); every symbol include/simhand_hip.h declares
SIGNATURES = {
    "simhand_abi_version": (_I, []),
    "simhand_half_format": (_I, []),
    "simhand_last_error": (C.c_char_p, []),
    "simhand_device_check": (_I, []),
    "simhand_route_counts": (_I, [_P]),
    "simhand_route_reset": (_I, []),
    "simhand_test_hooks_reset": (_I, []),
    "simhand_test_switch": (_I, [_I, _I]),
    "simhand_prof_enable": (_I, [_I]),
    "simhand_prof_set_classes": (_I, [C.c_uint32]),
    "simhand_prof_collect": (_I, [_P, _P, _P, _P]),
    "simhand_prof_records": (_I, [_I, _P, _P, _P, _P, _P]),
    "simhand_prof_reset": (_I, []),
    "simhand_pos_dist": (_I, [_P, _I, _I, _I, _P, _P, _P]),
    "simhand_neg_dist_workspace_bytes": (_S, [_I, _I]),
    "simhand_neg_dist": (_I, [_P, _I, _I, _I, _I, _I, _P, _P, _P, _S, _P]),
    "simhand_weights_from_dist": (_I, [_P, _L, _I, _P, _I, C.c_double, _F, _P, _P]),
    "simhand_ntxent_workspace_bytes":(_S, [C.POINTER(NtxentParams)]),
    "simhand_ntxent_fwd": (_I, [C.POINTER(NtxentParams), _P, _P, _P, _P, _P, _P, _P, _S, _P]),
    "simhand_ntxent_bwd": (_I, [C.POINTER(NtxentParams), _P, _P, _P, _P, _P, _P, _P, _P, _S, _P]),
    "simhand_ntxent_fwd_fused": (_I, [C.POINTER(NtxentParams), _P, _P, _I, _I, _P, _P, _P, _P, _P, _S, _P]),
    "simhand_ntxent_bwd_fused": (_I, [C.POINTER(NtxentParams), _P, _P, _I, _I, _P, _P, _P, _P, _P, _P, _S, _P]),
    "simhand_proj_postprocess_fwd": (_I, [_P, _I, _I, _P, _P, _P, _P, _P, _I, _I, _I, _P, _P]),
    "simhand_proj_postprocess_bwd": (_I, [_P, _I, _I, _P, _P, _P, _P, _P, _I, _I, _I, _P, _P, _P]),
    "simhand_proj_stats": (_I, [_P, _I, _I, _P, _P, _P]),
    "simhand_conv2d_fwd_stat_blocks": (_I, [C.POINTER(ConvDesc)]),
    "simhand_conv2d_fwd": (_I, [C.POINTER(ConvDesc), _P, _P, _P, _P, _P]),
    "simhand_conv2d_fwd_bnin_ok": (_I, [C.POINTER(ConvDesc)]),
    "simhand_conv2d_fwd_bnin": (_I, [C.POINTER(ConvDesc), _P, _P, _P, _P, _P, _P, _P, _P]),
    "simhand_conv2d_dgrad": (_I, [C.POINTER(ConvDesc), _P, _P, _P, _I, _P]),
    "simhand_conv2d_dgrad_masked_residual": (_I, [C.POINTER(ConvDesc), _P, _P, _P, _P, _P, _P]),
    "simhand_conv2d_wgrad_workspace_bytes": (_S, [C.POINTER(ConvDesc)]),
    "simhand_conv2d_wgrad": (_I, [C.POINTER(ConvDesc), _P, _P, _P, _P, _S, _P]),
    "simhand_test_conv1x1_set_rows": (_I, [_I, _I]),
    "simhand_stem_geometry": (_I, [_I, _I, _P, _P, _P, _P]),
    "simhand_stem_pad_input": (_I, [_P, _P, _I, _I, _I, _I, _P]),
    "simhand_stem_pack_weights": (_I, [_P, _P, _I, _P]),
    "simhand_stem_conv_fwd_stat_blocks": (_I, [_I, _I, _I, _I]),
    "simhand_stem_conv_fwd": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _P]),
    "simhand_stem_conv_wgrad_workspace_bytes": (_S, [_I, _I, _I, _I]),
    "simhand_stem_conv_wgrad": (_I, [_P, _P, _P, _P, _S, _I, _I, _I, _I, _P]),
    "simhand_conv2d_wgrad_splits": (_I, [C.POINTER(ConvDesc)]),
    "simhand_conv2d_wgrad_colsum": (_I, [C.POINTER(ConvDesc), _P, _P, _P, _P, _P, _S, _P]),
    "simhand_conv2d_wgrad_colsum_sums": (_I, [C.POINTER(ConvDesc), _P, _P, _P, _P, _P, _P, _S, _P]),
    "simhand_conv2d_wgrad_oihw": (_I, [C.POINTER(ConvDesc), _P, _P, _P, _I, _P, _S, _P]),
    "simhand_bn_apply_gram": (_I, [C.POINTER(ConvDesc), _P, _P, _P, _I, _P, _P, _P, _P, _S, _P]),
    "simhand_bn_apply_gram_sums": (_I, [C.POINTER(ConvDesc), _P, _P, _P, _I, _P, _P, _P, _P, _P, _S, _P]),
    "simhand_conv2d_wgrad_bnbwd": (_I, [C.POINTER(ConvDesc), _P, _P, _P, _P, _P, _P, _P, _P, _I, _P, _P, _I, _P, _S, _P]),
    "simhand_bn_bwd_coefs": (_I, [_P, _P, _P, _P, _P, _L, _I, _P, _P, _P, _P]),
    "simhand_test_wgrad_set_tr": (_I, [_I]),
    "simhand_test_wgrad_plain_kpm": (_I, [_I]),
    "simhand_test_wgrad_target_blocks": (_I, [_I, _I]),
    "simhand_test_wgrad3x3_enable": (_I, [_I]),
    "simhand_test_wgrad_dma_enable": (_I, [_I]),
    "simhand_test_bn_set_nt": (_I, [_I]),
    "simhand_test_igemm256_enable": (_I, [_I]),
    "simhand_nchw_f32_to_nhwc": (_I, [_P, _P, _I, _I, _I, _I, _I, _I, _P]),
    "simhand_oihw_f32_to_krsc": (_I, [_P, _P, _I, _I, _I, _I, _I, _I, _P]),
    "simhand_oihw_f32_to_crsk": (_I, [_P, _P, _I, _I, _I, _I, _I, _P]),
    "simhand_krsc_f32_to_oihw": (_I, [_P, _P, _I, _I, _I, _I, _I, _P]),
    "simhand_cast": (_I, [_P, _I, _P, _I, _L, _P]),
    "simhand_im2col_nchw_f32": (_I, [_P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P]),
    "simhand_bn_stat_blocks": (_I, [_L, _I]),
    "simhand_bn_partial_stats": (_I, [_P, _L, _I, _I, _P, _P]),
    "simhand_bn_finalize_workspace_bytes": (_S, [_I, _I]),
    "simhand_bn_finalize": (_I, [_P, _I, _L, _I, _P, _P, _P, _F, _F, _P, _P, _P, _P, _P, _P, _P, _P, _S, _P]),
    "simhand_bn_finalize_ticket": (_I, [_P, _I, _L, _I, _P, _P, _P, _F, _F, _P, _P, _P, _P, _P, _P, _P, _P, _S, _P, _P]),
    "simhand_bn_eval_params": (_I, [_P, _P, _P, _P, _F, _I, _P, _P, _P]),
    "simhand_bn_apply": (_I, [_P, _P, _P, _P, _I, _P, _P, _L, _I, _I, _P]),
    "simhand_bn_bwd_partial": (_I, [_P, _P, _P, _P, _P, _P, _P, _I, _L, _I, _I, _P, _P]),
    "simhand_bn_bwd_finalize": (_I, [_P, _I, _I, _P, _P, _P]),
    "simhand_bn_fold_workspace_bytes": (_S, [_I, _I]),
    "simhand_bn_fold_fwd": (_I, [_P, _I, _P, _P, _I, _I, _L, _P, _P, _F, _F, _P, _P, _P, _P, _P, _P, _P, _P, _P, _S, _P]),
    "simhand_bn_fold_bwd": (_I, [_P, _I, _P, _P, _P, _P, _P, _P, _P, _I, _I, _L, _P, _P, _P, _P, _P, _P, _P, _P, _I, _P, _S, _P]),
    "simhand_apply_relu_bitmask": (_I, [_P, _P, _P, _L, _I, _I, _P]),
    "simhand_bn_relu_maxpool_fwd": (_I, [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P]),
    "simhand_maxpool_bn_bwd_partial": (_I, [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P, _P]),
    "simhand_maxpool_bn_bwd_apply": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P]),
    "simhand_bn_bwd_finalize_raw_workspace_bytes": (_S, [_I, _I]),
    "simhand_bn_bwd_finalize_raw": (_I, [_P, _I, _I, _P, _P, _P, _P, _P, _S, _P]),
    "simhand_bn_bwd_finalize_raw_coefs": (_I, [_P, _I, _I, _P, _P, _P, _L, _P, _P, _P, _P, _S, _P]),
    "simhand_conv2d_dgrad_stat_blocks": (_I, [C.POINTER(ConvDesc), _I, _I, _I]),
    "simhand_test_conv3x3_c64_enable": (_I, [_I]),
    "simhand_test_conv3x3_r128_enable": (_I, [_I]),
    "simhand_test_stem_conv_route": (_I, [_I]),
    "simhand_test_igemm256_split_tail": (_I, [_I]),
    "simhand_test_igemm256_tile224": (_I, [_I]),
    "simhand_conv2d_dgrad_dysrc_ok": (_I, [C.POINTER(ConvDesc)]),
    "simhand_conv2d_fwd_chain_ok": (_I, [C.POINTER(ConvDesc)]),
    "simhand_test_conv1x1_chain_mask": (_I, [_I]),
    "simhand_conv2d_fwd_chain_stat_blocks": (_I, [C.POINTER(ConvDesc)]),
    "simhand_conv2d_fwd_bnact_chain": (_I, [C.POINTER(ConvDesc), _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "simhand_conv2d_dgrad_concat_ok": (_I, [C.POINTER(ConvDesc), _I]),
    "simhand_conv2d_fwd_bnact": (_I, [C.POINTER(ConvDesc), _P, _P, _P, _P, _P, _I, _P, _P, _P]),
    "simhand_conv2d_dgrad_ex": (_I, [C.POINTER(ConvDesc), _P, _P, _P, C.POINTER(DgradOpts), _P]),
    "simhand_conv2d_dgrad_fuse_pays": (_I, [C.POINTER(ConvDesc)]),
    "simhand_test_conv2d_dgrad_fuse_1x1": (_I, [_I]),
    "simhand_conv2d_dgrad_fused": (_I, [C.POINTER(ConvDesc), _P, _P, _P, _I, _P, _P, C.POINTER(BnBwdFuse), _P]),
    "simhand_bn_bwd_apply": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _P, _P, _L, _I, _I, _P]),
    "simhand_maxpool3x3s2_fwd": (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _P]),
    "simhand_maxpool3x3s2_bwd": (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _P]),
    "simhand_avgpool_fwd": (_I, [_P, _P, _I, _I, _I, _I, _P]),
    "simhand_subsample2": (_I, [_P, _P, _I, _I, _I, _I, _I, _P]),
    "simhand_scatter2_add": (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _P]),
    "simhand_avgpool_bwd": (_I, [_P, _P, _I, _I, _I, _I, _P]),
    "simhand_avgpool_bwd_masked": (_I, [_P, _P, _P, _I, _I, _I, _I, _P]),
    "simhand_colsum": (_I, [_P, _L, _I, _I, _P, _P, _P]),
    "simhand_sumsq_partial": (_I, [_P, _L, _P, _I, _P]),
    "simhand_opt_chunk_elems": (_I, []),
    "simhand_pack_chunk_elems": (_I, []),
    "simhand_pack_weights_multi": (_I, [_P, _P, _I, _P, _I, _I, _P]),
    "simhand_lars_adam_multi": (_I, [_P, _I, _P, _I, _P, _F, _F, _F, _F, _F, _I, _L, _P]),
    "simhand_lars_adam_multi_guarded": (_I, [_P, _I, _P, _I, _P, _F, _F, _F, _F, _F, _I, _L, _P, _P]),
    "simhand_fp8_state_floats": (_I, [_I]),
    "simhand_fp8_amax": (_I, [_P, _L, _I, _P, _P]),
    "simhand_fp8_scale_update": (_I, [_P, _P, _I, _F, _I, _P]),
    "simhand_fp8_quantize": (_I, [_P, _P, _L, _I, _P, _P, _P]),
    "simhand_fp8_pack_krsc": (_I, [_P, _P, _I, _I, _I, _I, _P, _P]),
    "simhand_conv2d_fwd_fp8_supported": (_I, [C.POINTER(ConvDesc)]),
    "simhand_conv2d_fwd_fp8_stat_blocks": (_I, [_P]),
    "simhand_conv2d_fwd_fp8_pays": (_I, [_P]),
    "simhand_conv2d_dgrad_fp8_pays": (_I, [_P]),
    "simhand_conv2d_wgrad_fp8_pays": (_I, [_P]),
    "simhand_conv2d_wgrad_fp8_workspace_bytes": (_S, [_P]),
    "simhand_conv2d_wgrad_fp8": (_I, [_P, _P, _P, _P, _P, _P, _P, _S, _P]),
    "simhand_bn_bwd_apply_fp8": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _P, _P, _P, _P, C.c_int64, _I, _P]),
    "simhand_bn_apply_fp8": (_I, [_P, _P, _P, _I, _P, _P, _P, _P, C.c_int64, _I, _P]),
    "simhand_conv2d_fwd_fp8": (_I, [C.POINTER(ConvDesc), _P, _P, _P, _P, _P, _P, _P]),
    "simhand_augment_workspace_bytes": (_S, [_I]),
    "simhand_augment_batch": (_I, [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P, _P, _P, _P, _S, _P]),
    "simhand_augment_workspace_bytes_ex": (_S, [_I, _I, _I, _I, _I]),
    "simhand_augment_batch_ex": (_I, [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P, _P, _P, _P, _S, _P]),
    "simhand_comm_unique_id": (_I, [_P]),
    "simhand_comm_init": (_I, [_P, _I, _I, C.POINTER(_P)]),
    "simhand_comm_destroy": (_I, [_P]),
    "simhand_comm_world": (_I, [_P, C.POINTER(_I), C.POINTER(_I)]),
    "simhand_comm_all_gather": (_I, [_P, _P, _P, _L, _I, _P]),
    "simhand_comm_all_reduce": (_I, [_P, _P, _P, _L, _I, _I, _P]),
    "simhand_lars_adam_step": (_I, [_P, _P, _P, _P, _L, _P, _P, _I, _F, _F, _F, _F, _F, _F, _F, _I, _I, _I, _P]),
}

_libs: dict = {}          # "bf16" / "f16" -> CDLL
_current = "bf16"         # which build load() hands out: the 16-bit storage format of the tensors the caller works with
_lock = threading.Lock()


def use_half(fmt: str) -> None:
    """Select the library build by its 16-bit storage format: "bf16" (default; BASELINE's benchmark dtype) or "f16" (IEEE fp16, the
    reference's precision=16).  Process-wide: one model / one format at a time (the fp32 parity mode works with either build)."""
    global _current, _device_ok
    if fmt not in ("bf16", "f16"):
        raise ValueError(fmt)
    if fmt != _current:
        _current = fmt
        _device_ok = False


def half_format() -> str:
    return _current


def half_dtype():
    import torch

    return torch.float16 if _current == "f16" else torch.bfloat16


def load() -> C.CDLL:
    """Load the in-tree shared library of the current 16-bit format (built by ``__graft_entry__.build()`` /
    ``make -C simhand_amd/csrc``).  Raises if it is absent."""
    lib = _libs.get(_current)
    if lib is not None:
        return lib
    with _lock:
        lib = _libs.get(_current)
        if lib is not None:
            return lib
        path = LIB_PATH_F16 if _current == "f16" else LIB_PATH
        if not os.path.exists(path):
            raise SimhandHipError(
                f"{path} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(there is no CPU fallback for the SiMHand hot path)")
        lib = C.CDLL(path)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)  # AttributeError if the ABI and this table disagree
            fn.restype = res
            fn.argtypes = args
        if lib.simhand_abi_version() != ABI_VERSION:
            raise SimhandHipError(f"{path} reports ABI version {lib.simhand_abi_version()}, this binding was written against {ABI_VERSION} "
                                  "(include/simhand_hip.h SH_ABI_VERSION): rebuild the library (`make -C simhand_amd/csrc`)")
        want = 1 if _current == "f16" else 0
        if lib.simhand_half_format() != want:
            raise SimhandHipError(f"{path} was built with 16-bit format {lib.simhand_half_format()}, expected {want}")
        _libs[_current] = lib
    return lib


def check(rc: int, what: str = "") -> None:
    if rc != 0:
        msg = load().simhand_last_error()
        raise SimhandHipError(f"{what} failed (rc={rc}): {msg.decode() if msg else 'unknown error'}")


_device_ok = False


def require_device() -> None:
    """Fail loudly unless a gfx950 GPU is usable (no CPU path exists)."""
    global _device_ok
    if _device_ok:
        return
    lib = load()
    check(lib.simhand_device_check(), "simhand_device_check")
    _device_ok = True
