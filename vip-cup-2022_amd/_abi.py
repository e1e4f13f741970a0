"""ctypes binding of include/vipcup_hip.h.  Loads the in-tree libvipcup_hip.so and fails loudly if it
is absent (there is deliberately no CPU fallback)."""
import ctypes as C
import os

# torch must load ITS libamdhip64 first: libvipcup_hip.so then binds to that same runtime instance (same
# SONAME), so device pointers and streams are shared.  Loaded the other way round the process ends up with
# two HIP runtimes and every launch fails with "no ROCm-capable device".
import torch  # noqa: F401  (side effect: HIP runtime)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("VIP_LIB_PATH") or os.path.join(_HERE, "libvipcup_hip.so")   # VIP_LIB_PATH: kernel A/B builds

_lib = None


class VipError(RuntimeError):
    pass


class ConvDesc(C.Structure):
    """mirror of vip_conv_desc (include/vipcup_hip.h)"""
    _fields_ = [(n, C.c_int) for n in (
        "B", "H", "W", "Cin", "Cout", "kh", "kw", "sh", "sw", "pt", "pl", "Ho", "Wo", "groups",
        "ldx", "cin_off", "ldy", "cout_off", "ldr", "res_off", "ldw", "act_pre", "act_post")]


class JpegDesc(C.Structure):
    """mirror of vip_jpeg_desc (include/vipcup_hip.h)"""
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("ncomp", C.c_int32),
                ("hsamp", C.c_int32 * 3), ("vsamp", C.c_int32 * 3),
                ("blocks_w", C.c_int32 * 3), ("blocks_h", C.c_int32 * 3),
                ("coef_off", C.c_int64 * 3), ("qt", (C.c_uint16 * 64) * 3), ("rgb_coded", C.c_int32)]


_vp, _i, _f, _sz = C.c_void_p, C.c_int, C.c_float, C.c_size_t

# name -> (restype, argtypes); every symbol include/vipcup_hip.h declares
SIGNATURES = {
    "vip_version": (_i, []),
    "vip_last_error": (C.c_char_p, []),
    "vip_conv2d_nhwc_f16": (_i, [_vp, _vp, _vp, _vp, _vp, C.POINTER(ConvDesc), _vp]),
    "vip_conv2d_gated_nhwc_f16": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, C.POINTER(ConvDesc), _vp]),
    "vip_conv2d_hilo_nhwc_f16": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, C.POINTER(ConvDesc), _vp]),
    "vip_gemm_bias_act_f16": (_i, [_vp, _vp, _vp, _vp, _vp] + [_i] * 9 + [_vp]),
    "vip_mlp_fused_supported": (_i, [_i, _i, _i, _i]),
    "vip_mlp_fused_f16": (_i, [_vp, _vp, _vp, _f] + [_vp] * 6 + [_i] * 9 + [_vp]),
    "vip_se_gate_f16": (_i, [_vp] * 6 + [_i] * 11 + [_vp]),
    "vip_gemm_split_f16": (_i, [_vp, _vp, _vp, _vp] + [_i] * 6 + [_vp]),
    "vip_gemm_split2_f16": (_i, [_vp, _vp, _vp, _vp] + [_i] * 5 + [_vp]),
    "vip_dwconv2d_nhwc_f16": (_i, [_vp, _vp, _vp, _vp] + [_i] * 11 + [_vp]),
    "vip_dwconv2d_pool_parts": (_i, [_i] * 8),
    "vip_dwconv2d_pool_nhwc_f16": (_i, [_vp, _vp, _vp, _vp, _vp] + [_i] * 12 + [_vp]),
    "vip_se_gate_pooled_f16": (_i, [_vp, _i] + [_vp] * 5 + [_i] * 10 + [_vp]),
    "vip_layernorm_f16": (_i, [_vp, _vp, _vp, _vp, _i, _i, _f, _vp]),
    "vip_pool2d_nhwc_f16": (_i, [_vp, _vp] + [_i] * 13 + [_vp]),
    "vip_global_avgpool_f16": (_i, [_vp, _vp, _i, _i, _i, _i, _vp]),
    "vip_global_avgpool_split_f16": (_i, [_vp, _vp, _i, _i, _i, _i, _vp]),
    "vip_gap_dense_f32": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "vip_experiments_built": (_i, []),
    "vip_gcvit_attn_block_supported": (_i, [_i, _i, _i]),
    "vip_gcvit_attn_block_f16": (_i, [_vp, _vp, _vp, _vp, _f, _vp, _i, _vp, _vp, _i, _vp, _vp, _vp] + [_i] * 6 + [_f, _vp]),
    "vip_mbconv_expand_dw_supported": (_i, [_i, _i, _i, _i]),
    "vip_mbconv_expand_dw_f16": (_i, [_vp] * 7 + [_i] * 14 + [_vp]),
    "vip_head_prob_f32": (_i, [_vp, _vp, _vp, _i, _i, _vp]),
    "vip_head_act_f32": (_i, [_vp, _vp, _i, _i, _i, _vp]),
    "vip_prob_to_score_f32": (_i, [_vp, _vp, _i, _i, _vp]),
    "vip_ensemble_mean_f32": (_i, [_vp, _vp, _i, _i, C.c_long, _vp]),
    "vip_gap_ln_dense_f32": (_i, [_vp, _vp, _vp, _f, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "vip_scale_add_act_f16": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "vip_scale_add_act2_f16": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "vip_scale_add_act3_f16": (_i, [_vp, _vp, _i, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "vip_mul_f16": (_i, [_vp, _vp, _vp, C.c_long] + [_i] * 7 + [_vp]),
    "vip_radix_combine_f16": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "vip_radix_combine2_f16": (_i, [_vp, _vp, _i, _vp, _i, _i, _i, _i, _vp]),
    "vip_window_attn_fwd_f16": (_i, [_vp, _vp, _vp, _vp] + [_i] * 7 + [_f, _vp]),
    "vip_mhsa_fwd_f16": (_i, [_vp, _vp, _i, _i, _i, _i, _f, _vp]),
    "vip_jpeg_probe_h": (_i, [_vp, _sz, _vp, _vp]),
    "vip_jpeg_entropy_decode_h": (_i, [_vp, _vp, _i, _vp, _vp, _sz, _vp, _i]),
    "vip_jpeg_idct_rgb_u8": (_i, [_vp, _vp, _i, _i, _vp, _vp, _i, _i, _vp]),
    "vip_bicubic_table_f32": (_i, [_vp]),
    "vip_resize_bicubic_norm_f16": (_i, [_vp, _vp, _vp, _i, _i, _i, _vp, _i, _i, _i, _vp]),
    "vip_tta_augment_f16": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "vip_vit_tokens_f16": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _vp]),
    # STRICT precision path (fp32 storage, fp32 arithmetic): csrc/strict_conv.hip, csrc/strict_ops.hip
    "vip_conv2d_nhwc_s32": (_i, [_vp, _vp, _vp, _vp, _vp, C.POINTER(ConvDesc), _vp]),
    "vip_conv2d_nhwc_s32x": (_i, [_vp, _vp, _i, _vp, _vp, _vp, C.POINTER(ConvDesc), _vp]),
    "vip_conv2d_nhwc_s32x2": (_i, [_vp, _vp, _i, _vp, _vp, _vp, C.POINTER(ConvDesc), _vp]),
    "vip_dwconv2d_nhwc_s32": (_i, [_vp, _vp, _vp, _vp] + [_i] * 11 + [_vp]),
    "vip_layernorm_s32": (_i, [_vp, _vp, _vp, _vp, _i, _i, _f, _vp]),
    "vip_pool2d_nhwc_s32": (_i, [_vp, _vp] + [_i] * 13 + [_vp]),
    "vip_global_avgpool_s32": (_i, [_vp, _vp, _i, _i, _i, _i, _vp]),
    "vip_scale_add_act_s32": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "vip_radix_combine_s32": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "vip_mul_s32": (_i, [_vp, _vp, _vp, C.c_long] + [_i] * 7 + [_vp]),
    "vip_vit_tokens_s32": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _vp]),
    "vip_gap_ln_dense_s32": (_i, [_vp, _vp, _vp, _f, _vp, _vp, _vp, _i, _i, _i, _i, C.c_long, _i, _vp]),
    "vip_window_attn_fwd_s32": (_i, [_vp, _vp, _vp, _vp] + [_i] * 7 + [_f, _vp]),
    "vip_mhsa_fwd_s32": (_i, [_vp, _vp, _i, _i, _i, _i, _f, _vp]),
    "vip_resize_bicubic_norm_s32": (_i, [_vp, _vp, _vp, _i, _i, _i, _vp, _i, _i, _i, _vp]),
    "vip_tta_augment_s32": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    # STRICT precision path, packed (hi, lo) fp16 storage: csrc/conv_h2.hip, csrc/strict_ops.hip
    "vip_pack_h2": (_i, [_vp, _vp, C.c_long, _vp, _vp]),
    "vip_unpack_h2": (_i, [_vp, _vp, C.c_long, _vp]),
    "vip_conv2d_nhwc_h2": (_i, [_vp, _vp, _vp, _vp, _vp, C.POINTER(ConvDesc), _f, _vp, _vp]),
    "vip_conv2d_gated_nhwc_h2": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, C.POINTER(ConvDesc), _f, _vp, _vp]),
    "vip_conv2d_kernel_name_h2": (_i, [C.POINTER(ConvDesc), _i, C.c_char_p, _sz]),
    "vip_dwconv2d_nhwc_h2": (_i, [_vp, _vp, _vp, _vp] + [_i] * 11 + [_vp, _vp]),
    "vip_mlp_fused_supported_h2": (_i, [_i] * 4),
    "vip_mlp_fused_h2": (_i, [_vp, _vp, _vp, _f, _vp, _vp, _f, _vp, _vp, _f, _vp, _vp] + [_i] * 9 + [_vp, _vp]),
    "vip_dw_filter_quad_major": (_i, [_vp, _vp, _i, _i, _vp]),
    "vip_dwconv2d_s1_supported_h2": (_i, [_i] * 7),
    "vip_dwconv2d_s1_h2": (_i, [_vp, _vp, _vp, _vp] + [_i] * 10 + [_vp, _vp]),
    "vip_dwconv2d_s1_pool_parts_h2": (_i, [_i] * 7),
    "vip_dwconv2d_s1_pool_h2": (_i, [_vp, _vp, _vp, _vp, _vp] + [_i] * 11 + [_vp, _vp]),
    "vip_se_gate_pooled_h2": (_i, [_vp, _i, _vp, _vp, _f, _vp, _vp, _f, _vp] + [_i] * 9 + [_vp, _vp]),
    "vip_se_gate_h2": (_i, [_vp, _vp, _vp, _f, _vp, _vp, _f, _vp] + [_i] * 10 + [_vp, _vp]),
    "vip_layernorm_h2": (_i, [_vp, _vp, _vp, _vp, _i, _i, _f, _vp, _vp]),
    "vip_pool2d_nhwc_h2": (_i, [_vp, _vp] + [_i] * 13 + [_vp, _vp]),
    "vip_global_avgpool_h2": (_i, [_vp, _vp, _i, _i, _i, _i, _vp, _vp]),
    "vip_scale_add_act_h2": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _vp]),
    "vip_radix_combine_h2": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp]),
    "vip_mul_h2": (_i, [_vp, _vp, _vp, C.c_long] + [_i] * 7 + [_vp, _vp]),
    "vip_vit_tokens_h2": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _vp, _vp]),
    "vip_gap_ln_dense_h2": (_i, [_vp, _vp, _vp, _f, _vp, _vp, _vp, _i, _i, _i, _i, C.c_long, _i, _vp]),
    "vip_window_attn_fwd_h2": (_i, [_vp, _vp, _vp, _vp] + [_i] * 7 + [_f, _vp, _vp]),
    "vip_mhsa_fwd_h2": (_i, [_vp, _vp, _i, _i, _i, _i, _f, _vp, _vp]),
    "vip_conv2d_kernel_name": (_i, [C.POINTER(ConvDesc), _i, _i, _i, C.c_char_p, _sz]),
    "vip_workspace_bytes": (_sz, [_i, C.POINTER(C.c_int64), _i]),
    "vip_microbench_copy": (_i, [_vp, _vp, _sz, _vp]),
    "vip_microbench_copy_variant": (_i, [_vp, _vp, _sz, _i, _vp]),
    "vip_microbench_mfma_f16": (_i, [_vp, _i, C.POINTER(C.c_double), _vp]),
}


def lib():
    """Return the loaded library (loading it on first use)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise VipError(
                f"{LIB_PATH} not found: build it with `python vip-cup-2022_amd/build.py` "
                "(or __graft_entry__.build()); vipcup_amd has no CPU fallback")
        l = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(l, name)  # AttributeError if the .so does not export a declared symbol
            fn.restype = res
            fn.argtypes = args
        _lib = l
    return _lib


def check(status, what):
    if status != 0:
        msg = lib().vip_last_error().decode("utf-8", "replace")
        raise VipError(f"{what} failed with vip_status {status}: {msg}")
