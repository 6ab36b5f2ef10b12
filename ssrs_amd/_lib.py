"""ctypes binding of libmfnet_hip.so (C ABI declared in include/mfnet_hip.h).

The product path has NO fallback: if the library is missing or a call fails, an exception is raised.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import c_char_p, c_double, c_float, c_int, c_int64, c_void_p

# torch must be loaded FIRST: it ships its own libamdhip64.so.7 and owns the device memory and streams
# this library works on. Loading libmfnet_hip.so before torch would bind the process to the system HIP
# runtime instead and torch's allocations would belong to another runtime ("no ROCm-capable device").
import torch  # noqa: F401

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MFNET_HIP_LIB") or os.path.join(_HERE, "libmfnet_hip.so")     # override: A/B runs of two builds

F32, BF16 = 0, 1


class MfnetError(RuntimeError):
    pass


_lib = None

P, I64, I, F, D = c_void_p, c_int64, c_int, c_float, c_double

# name -> argtypes (all return int status); mirrors include/mfnet_hip.h
_SIGS = {
    "mfnet_layernorm_fwd": [P, I, I64, P, P, P, I, I64, P, P, I64, I64, F, P],
    "mfnet_layernorm_fwd_lora": [P, I, I64, P, P, P, I64, P, P, I64, I64, F, P, I64, I, I, P, P, P, P, P],
    "mfnet_lora_down_mix": [P, I64, I64, I64, P, I64, I, I, P, P, P, P, P],
    "mfnet_layernorm_bwd": [P, I, I64, P, I, I64, P, P, P, P, P, P, I64, P, P, I64, I64, P],
    "mfnet_gemm_bf16": [P, I64, P, I64, I64, I64, I64, P, P, I64, P, I64, I64, I, I, F, I, P, I64, P, I64,
                        P, I64, I64, P, I, I64, I, P],
    "mfnet_gemm_bf16_tiled": [P, I64, P, I64, I64, I64, I64, P, P, I64, P, I64, I64, I, I, F, I, P, I64, P, I64,
                              P, I64, I64, P, I, I64, I, I, P],
    "mfnet_gemm_bf16_grouped": [P, I64, P, I64, I64, I64, I64, I64, P, I64, I64, I, I, P, I64, P, I64, P, I64, P, I, I64, P],
    "mfnet_skinny_nt": [P, I64, P, I64, I64, I64, I, P, I64, I64, I64, I, F, P],
    "mfnet_skinny_tn": [P, I64, P, I64, P, I64, I64, I, I, F, I, I64, P, I64, P],
    "mfnet_cast": [P, I, P, I, I64, P],
    "mfnet_im2col_patch": [P, P, I, P, I64, I64, I64, I64, P],
    "mfnet_im2col3x3": [P, P, I64, I64, I64, I64, P],
    "mfnet_col2im3x3": [P, P, I64, I64, I64, I64, P],
    "mfnet_lora_mix_fwd": [P, P, P, P, I64, I, P],
    "mfnet_lora_mix_bwd": [P, I, I64, P, P, P, P, P, I64, I, P],
    "mfnet_lora_side": [P, I64, I64, I64, I, P, I64, I, P, P, I64, I64, I, P, P],
    "mfnet_attn_fwd": [P, P, P, P, P, I64, I64, I64, I64, I64, I64, I64, I64, I64, F, P],
    "mfnet_attn_bwd": [P, P, P, P, P, P, P, P, I64, I64, I64, I64, I64, I64, I64, I64, I64, F, P, I64, P],
    "mfnet_ce_loss_fwd": [P, P, P, P, P, I64, I64, I64, I64, P],
    "mfnet_ce_loss_bwd": [P, P, P, P, P, F, I64, I64, I64, I64, P],
    "mfnet_sgd_flat": [P, P, P, I64, F, F, F, F, P],
    "mfnet_tile_accumulate": [P, P, P, I64, I64, I64, I64, I64, I64, I, P],
    "mfnet_argmax_chw": [P, P, I64, I64, P],
    "mfnet_tile_accumulate_f64": [P, P, I, P, I64, I64, I64, I64, I64, I64, I, P],
    "mfnet_argmax_chw_f64": [P, P, I64, I64, P],
    "mfnet_minmax": [P, P, I64, P],
    "mfnet_scale_shift": [P, P, P, F, I64, I, P],
    "mfnet_adapter_mix": [P, P, P, P, P, P, I64, I64, P],
    "mfnet_adapter_mix_bwd": [P, P, P, P, P, P, P, I64, I64, P],
    "mfnet_gelu_fwd": [P, P, I64, P],
    "mfnet_gelu_bwd": [P, P, P, I64, P],
    "mfnet_add_bf16": [P, P, P, I64, P],
    "mfnet_bias_grad": [P, I64, P, I64, I64, P],
    "mfnet_transpose_bf16": [P, I64, P, I64, I64, I64, P],
    "mfnet_reduce_splits": [P, P, I64, I, F, I, P],
    "mfnet_gather_segments": [P, I, I, I, P],
    "mfnet_prepare_tiles": [P, P, P, P, P, P, P, I64, I64, P, P, P],
    "mfnet_gemm_tn_bf16": [P, I64, P, I64, I64, I64, I64, P, I64, I, P, P],
    "mfnet_conv3x3_bf16": [P, I64, P, I64, I64, I64, I64, I64, I64, P, I, I64, P],
    "mfnet_gemm_tn_conv3x3": [P, I64, P, I64, I64, I64, I64, I64, I64, P, I64, I, P, P],
    "mfnet_gemm_tn_bf16_grouped": [P, I64, P, I64, I64, I64, I64, P, I64, I, I, P, P],
    "mfnet_gemm_tn_batch": [P, I64, P],
    "mfnet_lora_side_batch": [P, I64, I, P],
    "mfnet_dwconv_bwd_weight_batch": [P, I64, P],
    # ---- FPN / SE-fusion / UNetFormer decoder (csrc/decoder.hip)
    "mfnet_colsum": [P, I, P, P, I64, I64, I64, P, P, P],
    "mfnet_colsum_static": [P, I, P, P, I64, I64, I64, P],
    "mfnet_coldot": [P, I, P, I, P, I64, I64, I64, P, P, P],
    "mfnet_layernorm_bwd_weights": [P, I, P, I, P, P, P, P, I64, I64, P, P, P],
    "mfnet_bn_finalize": [P, P, P, P, P, P, P, I64, I64, F, F, I, P],
    "mfnet_bn_train_apply": [P, I, P, P, P, P, P, P, P, F, F, P, P, I, P, I64, P, I, P, I, I64, I64, P],
    "mfnet_bn_apply": [P, I, P, P, P, P, I, P, I64, P, I, P, I, I64, I64, P],
    "mfnet_bn_bwd_reduce": [P, I, P, I, P, P, P, P, I, P, I64, P, P, I64, I64, P, P, P],
    "mfnet_bn_bwd_apply": [P, I, P, I, P, P, P, P, I, P, I64, P, P, I, P, I, I64, I64, P],
    "mfnet_act": [P, I, P, I, I, I64, P],
    "mfnet_act_bwd": [P, I, P, I, P, I, I, I64, P],
    "mfnet_dwconv_fwd": [P, P, P, P, I64, I64, I64, I64, I64, I, P],
    "mfnet_dwconv_bwd_data": [P, I, P, P, I64, I64, I64, I64, I64, I, P],
    "mfnet_dwconv_bwd_weight": [P, I, P, P, P, I64, I64, I64, I64, I64, I, I, P],
    "mfnet_gla_attn_fwd": [P, P, P, I64, I64, I64, I64, I64, I64, F, P],
    "mfnet_gla_attn_bwd": [P, P, P, P, P, I64, I64, I64, I64, I64, I64, F, P],
    "mfnet_gla_pool_fwd": [P, P, P, P, I64, I64, I64, I64, P],
    "mfnet_gla_pool_bwd": [P, P, I64, I64, I64, I64, P],
    "mfnet_upfuse_fwd": [P, P, I, I, P, F, P, I, I64, I64, I64, I64, P],
    "mfnet_upfuse_bwd": [P, P, P, I, I, P, F, P, P, I, P, P, I64, I64, I64, I64, P],
    "mfnet_up_final_fwd": [P, I64, P, I64, I64, I64, I64, I64, P],
    "mfnet_up_final_bwd": [P, P, I64, I64, I64, I64, I64, I64, P],
    "mfnet_maxpool2_fwd": [P, P, P, I64, I64, I64, I64, P],
    "mfnet_maxpool2_bwd": [P, I, P, P, I64, I64, I64, I64, I, P],
    "mfnet_se_gate_fwd": [P, I64, P, P, P, P, I, P, P, I64, I64, I64, P],
    "mfnet_se_gate_bwd": [P, P, I64, P, P, I, P, P, P, P, P, P, P, I64, I64, I64, P],
    "mfnet_se_scale_sum": [P, I, P, I, P, P, P, I, I64, I64, I64, P],
    "mfnet_se_bwd_apply": [P, I, P, P, P, P, P, I, P, I, I64, I64, I64, P],
    "mfnet_frh_gate_fwd": [P, P, P, P, I64, I64, I64, P],
    "mfnet_frh_gate_bwd": [P, P, P, P, P, P, P, I64, I64, I64, P],
    "mfnet_add3": [P, I, P, I, P, I, P, I, I64, P],
    # ---- raster inference pre / post (csrc/infer.hip)
    "mfnet_normalize_rgb_u8": [P, P, I64, I64, F, P, P, P],
    "mfnet_extract_tiles": [P, P, P, P, P, I64, I64, I64, I64, I64, I64, P],
    "mfnet_tile_minmax_norm": [P, I64, I64, I, P],
    "mfnet_jitter_tiles": [P, P, I64, I64, P],
    "mfnet_tile_prob_accumulate": [P, P, P, P, I64, I64, I64, I64, I64, I64, I64, P],
    "mfnet_blend_finish": [P, P, P, P, F, I64, P],
    "mfnet_normalize_raster": [P, I, I64, I64, I64, I, I, F, F, P, P, I, D, P, P, P],
    "mfnet_tile_prob_mask": [P, P, P, P, I64, I64, I64, I64, I64, F, P],
    "mfnet_confusion": [P, P, P, I64, I64, P],
    # ---- fp32-accurate forward path (csrc/fp32.hip)
    "mfnet_gemm_f32": [P, I64, P, I64, I64, I64, I64, P, F, I, P, I64, I64, P, I64, I, P],
    "mfnet_attn_f32": [P, P, P, P, P, I64, I64, I64, I64, I64, I64, F, P],
    "mfnet_im2col_patch_f32": [P, P, I, P, I64, I64, I64, I64, P],
    "mfnet_im2col3x3_f32": [P, P, I64, I64, I64, I64, P],
    "mfnet_gla_attn_fwd_f32": [P, P, P, I64, I64, I64, I64, I64, I64, F, P],
    # ---- fp32 training: backward twins with fp32 operands / results
    "mfnet_attn_f32_bwd": [P, P, P, P, P, P, P, I64, I64, I64, I64, I64, I64, F, P, I64, P],
    "mfnet_gla_attn_bwd_f32": [P, P, P, P, P, I64, I64, I64, I64, I64, I64, F, P],
    "mfnet_up_final_bwd_f32": [P, P, I64, I64, I64, I64, I64, I64, P],
    "mfnet_adapter_mix_bwd_f32": [P, P, P, P, P, P, I64, I64, P],
}


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise MfnetError(
                f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(or `make -C ssrs_amd/csrc`). There is no CPU / eager fallback for the MFNet hot path.")
        L = ctypes.CDLL(LIB_PATH)
        L.mfnet_last_error.restype = c_char_p
        L.mfnet_last_error.argtypes = []
        L.mfnet_version.restype = c_int
        L.mfnet_version.argtypes = []
        L.mfnet_skinny_tn_workspace_floats.restype = c_int64
        L.mfnet_skinny_tn_workspace_floats.argtypes = [c_int64, c_int64, c_int, c_int]
        L.mfnet_colreduce_workspace_floats.restype = c_int64
        L.mfnet_colreduce_workspace_floats.argtypes = [c_int64, c_int64, c_int64]
        L.mfnet_lora_side_splits.restype = c_int64
        L.mfnet_lora_side_splits.argtypes = [c_int64, c_int64]
        L.mfnet_gemm_tn_splits.restype = c_int64
        L.mfnet_gemm_tn_splits.argtypes = [c_int64, c_int64, c_int64]
        L.mfnet_attn_bwd_workspace_floats.restype = c_int64
        L.mfnet_attn_bwd_workspace_floats.argtypes = [c_int64, c_int64, c_int64, c_int64, c_int64]
        L.mfnet_attn_f32_bwd_workspace_floats.restype = c_int64
        L.mfnet_attn_f32_bwd_workspace_floats.argtypes = [c_int64, c_int64, c_int64, c_int64, c_int64]
        for name, sig in _SIGS.items():
            fn = getattr(L, name, None)
            if fn is None:
                raise MfnetError(f"libmfnet_hip.so does not export {name}")
            fn.restype = c_int
            fn.argtypes = sig
        _lib = L
    return _lib


def exported_symbols():
    return ["mfnet_version", "mfnet_last_error", "mfnet_skinny_tn_workspace_floats", "mfnet_gemm_tn_splits", "mfnet_colreduce_workspace_floats",
            "mfnet_attn_bwd_workspace_floats", "mfnet_lora_side_splits", "mfnet_attn_f32_bwd_workspace_floats"] + list(_SIGS)


def call(name: str, *args):
    L = lib()
    rc = getattr(L, name)(*args)
    if rc != 0:
        raise MfnetError(f"{name} failed ({rc}): {L.mfnet_last_error().decode()}")
