"""ctypes binding of libuenc_hip.so (C ABI: include/uenc.h).

Every entry point takes raw device pointers, sizes / strides and a hipStream_t; it allocates
nothing and never synchronises.  Return convention: 0 ok, <0 invalid argument (nothing launched),
>0 hipError_t.  `check()` turns a non-zero code into a Python exception.
"""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libuenc_hip.so")

F32, BF16 = 0, 1
EPI_NONE, EPI_GELU, EPI_RELU, EPI_RESIDUAL, EPI_MUL_DGELU, EPI_MUL_DRELU = range(6)

c_p, c_i, c_l, c_f, c_u = ctypes.c_void_p, ctypes.c_int, ctypes.c_long, ctypes.c_float, ctypes.c_uint

# name -> argtypes; every function returns int unless noted
_SIGNATURES = {
    "uenc_version": [],
    "uenc_prof_enable": [c_i],
    "uenc_prof_collect": [c_i, c_p, c_p, c_p],
    "uenc_prof_collect_bytes": [c_i, c_p],
    "uenc_prof_next_bytes": [ctypes.c_double],
    "uenc_cast_f32_bf16": [c_p, c_p, c_l, c_p],
    "uenc_cast_transpose_f32_bf16": [c_p, c_p, c_i, c_i, c_p],
    "uenc_cast_multi": [c_p, c_i, c_l, c_p],
    "uenc_upsample_bilinear": [c_p, c_p, c_l, c_i, c_i, c_i, c_i, c_p],
    "uenc_attn_mask": [c_p, c_p, c_l, c_i, c_i, c_i, c_i, c_p],
    "uenc_gemm_nt": [c_p, c_i, c_l, c_p, c_l, c_p, c_i, c_l, c_i, c_i, c_i, c_p, c_i, c_p, c_l, c_p, c_l, c_f, c_i, c_i, c_p],
    "uenc_gemm_nt_scaled": [c_p, c_i, c_l, c_p, c_l, c_p, c_i, c_l, c_i, c_i, c_i, c_p, c_i, c_p, c_l, c_p, c_l, c_f, c_p, c_i, c_p],
    "uenc_gemm_nt_ln": [c_p, c_i, c_l, c_p, c_l, c_p, c_l, c_i, c_i, c_i, c_p, c_p, c_l, c_p, c_p, c_f, c_p, c_p, c_p, c_p],
    "uenc_groupnorm_tokens_scratch_bytes": [c_i, c_i, c_i, c_i],
    "uenc_groupnorm_tokens_fwd": [c_p, c_i, c_p, c_p, c_p, c_i, c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_f, c_i, c_p],
    "uenc_groupnorm_tokens_bwd": [c_p, c_i, c_p, c_i, c_p, c_p, c_p, c_p, c_i, c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_p],
    "uenc_upsample_bilinear_tokens_bwd": [c_p, c_i, c_p, c_i, c_i, c_i, c_i, c_i, c_i, c_p],
    "uenc_im2col3x3": [c_p, c_p, c_i, c_i, c_i, c_i, c_p],
    "uenc_col2im3x3": [c_p, c_p, c_i, c_i, c_i, c_i, c_p],
    "uenc_add_cast_bf16": [c_p, c_p, c_p, c_l, c_l, c_p],
    "uenc_msda_prep_fwd": [c_p, c_l, c_p, c_i, c_p, c_p, c_p, c_l, c_i, c_i, c_i, c_i, c_p],
    "uenc_msda_prep_bwd": [c_p, c_p, c_p, c_p, c_p, c_l, c_l, c_i, c_i, c_i, c_i, c_p],
    "uenc_segment_colsum": [c_p, c_l, c_i, c_p, c_i, c_l, c_i, c_p, c_p],
    "uenc_dropout_bf16": [c_p, c_p, c_l, c_u, c_f, c_p],
    "uenc_gemm_nt_splits": [c_i, c_i],
    "uenc_gemm_nt_partials": [c_p, c_i, c_l, c_p, c_l, c_p, c_l, c_l, c_i, c_i, c_i, c_f, c_i, c_p],
    "uenc_gemm_nt_batched": [c_p, c_i, c_l, c_l, c_p, c_l, c_l, c_p, c_i, c_l, c_l, c_i, c_i, c_i, c_i, c_f, c_i, c_i, c_p],
    "uenc_gemm_tn": [c_p, c_i, c_l, c_p, c_i, c_l, c_p, c_l, c_p, c_i, c_i, c_i, c_i, c_p],
    "uenc_gemm_tn_scaled": [c_p, c_i, c_l, c_p, c_i, c_l, c_p, c_l, c_p, c_i, c_i, c_i, c_i, c_f, c_p],
    "uenc_gemm_tn_grouped": [c_p, c_i, c_i, c_i, ctypes.c_double, c_p],
    "uenc_gemm_tn_grouped_small": [c_p, c_i, c_i, ctypes.c_double, c_p],
    "uenc_layernorm_fwd": [c_p, c_i, c_p, c_i, c_p, c_p, c_p, c_p, c_i, c_p, c_l, c_i, c_f, c_p, c_p],
    "uenc_layernorm_bwd": [c_p, c_i, c_p, c_i, c_p, c_p, c_p, c_p, c_i, c_p, c_p, c_l, c_i, c_p, c_p, c_i, c_p],
    "uenc_layernorm_bwd_blocks": [c_l, c_i],
    "uenc_ln_param_grouped": [c_p, c_i, c_i, c_p],
    "uenc_window_attn_bwd_groups": [c_i, c_i, c_i, c_i, c_i],
    "uenc_window_attn_dtable_grouped": [c_p, c_i, c_i, c_p],
    "uenc_msdeform_attn_fwd": [c_p, c_i, c_p, c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_p],
    "uenc_msdeform_attn_fwd_tiled": [c_p, c_i, c_p, c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_p, c_p],
    "uenc_msdeform_attn_bwd": [c_p, c_i, c_p, c_p, c_p, c_p, c_p, c_i, c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_p, c_p, c_l, c_p],
    "uenc_msdeform_attn_bwd_workspace_bytes": [c_p, c_i, c_i, c_i, c_i, c_i, c_i],
    "uenc_msdeform_attn_fused_fwd": [c_p, c_i, c_p, c_p, c_p, c_l, c_p, c_i, c_p, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_p],
    "uenc_msdeform_attn_fused_fwd_tiled": [c_p, c_i, c_p, c_p, c_p, c_l, c_p, c_i, c_p, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_p, c_p],
    "uenc_msdeform_attn_fused_bwd": [c_p, c_i, c_p, c_p, c_p, c_l, c_p, c_i, c_p, c_i, c_p, c_p, c_l, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_p, c_p, c_l, c_p],
    "uenc_mha_fwd_workspace_floats": [c_i, c_i, c_i, c_i],
    "uenc_mha_fwd": [c_p, c_l, c_l, c_p, c_l, c_l, c_p, c_l, c_l, c_p, c_l, c_p, c_l, c_l, c_p, c_p, c_i, c_i, c_i, c_i, c_f, c_f, c_u, c_p],
    "uenc_mha_bwd": [c_p, c_l, c_l, c_p, c_l, c_l, c_p, c_l, c_l, c_p, c_l, c_p, c_l, c_l, c_p, c_p, c_l, c_l, c_p, c_l, c_l,
                     c_p, c_l, c_l, c_p, c_l, c_l, c_i, c_i, c_i, c_i, c_f, c_f, c_u, c_p],
    "uenc_window_attn_np": [c_i],
    "uenc_relpos_expand": [c_p, c_p, c_p, c_i, c_i, c_p],
    "uenc_relpos_expand_grouped": [c_p, c_i, c_i, c_p],
    "uenc_window_attn_fwd": [c_p, c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_f, c_p],
    "uenc_postproc_semantic": [c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_p],
    "uenc_postproc_panoptic_stats": [c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_p],
    "uenc_postproc_panoptic_label": [c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_p],
    "uenc_patch_merge_ln_fwd": [c_p, c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_f, c_p],
    "uenc_patch_merge_ln_bwd": [c_p, c_i, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_p],
    "uenc_im2col3x3_s2": [c_p, c_p, c_i, c_i, c_i, c_i, c_p],
    "uenc_col2im3x3_s2": [c_p, c_p, c_i, c_i, c_i, c_i, c_p],
    "uenc_na2d_fwd": [c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_i, c_f, c_p],
    "uenc_na2d_bwd": [c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_i, c_f, c_p],
    "uenc_window_attn_bwd_ws_floats": [c_i, c_i, c_i, c_i, c_i],
    # fp32 exact mode (csrc/exact.hip)
    "uenc_gemm_nt_f32": [c_p, c_l, c_p, c_l, c_p, c_l, c_i, c_i, c_i, c_p, c_i, c_p, c_l, c_p, c_l, c_f, c_i, c_p],
    "uenc_gemm_tn_f32": [c_p, c_l, c_p, c_l, c_p, c_l, c_p, c_i, c_i, c_i, c_p],
    "uenc_window_attn_f32_fwd": [c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_f, c_p],
    "uenc_window_attn_f32_bwd": [c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_f, c_p],
    "uenc_mha_f32_fwd": [c_p, c_l, c_l, c_p, c_l, c_l, c_p, c_l, c_l, c_p, c_l, c_p, c_l, c_l, c_p, c_i, c_i, c_i, c_i, c_f, c_f, c_u, c_p],
    "uenc_mha_f32_bwd": [c_p, c_l, c_l, c_p, c_l, c_l, c_p, c_l, c_l, c_p, c_l, c_p, c_l, c_l, c_p, c_p, c_l, c_l, c_p, c_l, c_l,
                         c_p, c_l, c_l, c_p, c_l, c_l, c_p, c_i, c_i, c_i, c_i, c_f, c_f, c_u, c_p],
    "uenc_window_attn_bwd": [c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_f, c_i, c_p],
}


class UencError(RuntimeError):
    pass


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(or `make -C uni-encoder-code_amd/csrc`). The HIP library is mandatory; there is no fallback path.")
    lib = ctypes.CDLL(LIB_PATH)
    for name, argtypes in _SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if the ABI and the header drift apart
        fn.argtypes = argtypes
        fn.restype = c_i
    lib.uenc_window_attn_bwd_ws_floats.restype = c_l
    lib.uenc_mha_fwd_workspace_floats.restype = c_l
    lib.uenc_msdeform_attn_bwd_workspace_bytes.restype = c_l
    lib.uenc_groupnorm_tokens_scratch_bytes.restype = c_l
    lib.uenc_arch.restype = ctypes.c_char_p
    lib.uenc_arch.argtypes = []
    return lib


lib = _load()


def exported_symbols():
    return sorted(list(_SIGNATURES) + ["uenc_arch"])


def check(code: int, what: str = "uenc"):
    if code == 0:
        return
    if code < 0:
        raise UencError(f"{what}: invalid argument (shape / alignment / dtype), nothing was launched")
    raise UencError(f"{what}: hipError_t {code}")


def stream_ptr():
    return torch.cuda.current_stream().cuda_stream


def dt(t: torch.Tensor) -> int:
    if t.dtype == torch.float32:
        return F32
    if t.dtype == torch.bfloat16:
        return BF16
    raise UencError(f"unsupported dtype {t.dtype}")


def ptr(t):
    return 0 if t is None else t.data_ptr()
