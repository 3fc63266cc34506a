"""Thin launch wrappers: torch tensors in, C-ABI calls out.  No autograd here (see `ops.py`).

Every wrapper validates what the kernel's grid assumes (device, dtype, inner stride 1) before
handing raw pointers to the library; the library re-checks alignment and sizes and refuses
(-1) rather than launching on a bad shape.
"""
from typing import Optional

import torch

from . import capi
from .capi import BF16, F32, check, dt, lib, ptr, stream_ptr

EPI_NONE, EPI_GELU, EPI_RELU, EPI_RESIDUAL, EPI_MUL_DGELU, EPI_MUL_DRELU = (
    capi.EPI_NONE, capi.EPI_GELU, capi.EPI_RELU, capi.EPI_RESIDUAL, capi.EPI_MUL_DGELU, capi.EPI_MUL_DRELU)


def _mat(t: torch.Tensor, name: str):
    if not t.is_cuda:
        raise capi.UencError(f"{name} must live on the GPU")
    if t.dim() != 2 or t.stride(1) != 1:
        raise capi.UencError(f"{name} must be 2-D with unit inner stride, got {tuple(t.shape)} / {t.stride()}")
    return t


def cast_bf16(src: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """fp32 -> bf16 copy (weights).  numel must be a multiple of 8, else falls to the padded path."""
    assert src.dtype == torch.float32 and src.is_contiguous() and src.is_cuda
    if out is None:
        out = torch.empty(src.shape, dtype=torch.bfloat16, device=src.device)
    n = src.numel()
    if n % 8 == 0 and src.data_ptr() % 16 == 0:
        check(lib.uenc_cast_f32_bf16(src.data_ptr(), out.data_ptr(), n, stream_ptr()), "cast_f32_bf16")
    else:  # tiny odd-sized vectors (biases of odd length): treated as a 1 x n transpose
        check(lib.uenc_cast_transpose_f32_bf16(src.data_ptr(), out.data_ptr(), 1, n, stream_ptr()), "cast_f32_bf16")
    return out


def cast_transpose_bf16(src: torch.Tensor) -> torch.Tensor:
    """(R, C) fp32 -> (C, R) bf16."""
    assert src.dtype == torch.float32 and src.dim() == 2 and src.is_contiguous() and src.is_cuda
    R, C = src.shape
    out = torch.empty((C, R), dtype=torch.bfloat16, device=src.device)
    check(lib.uenc_cast_transpose_f32_bf16(src.data_ptr(), out.data_ptr(), R, C, stream_ptr()), "cast_transpose")
    return out


def gemm_nt(a: torch.Tensor, w: torch.Tensor, *, bias: Optional[torch.Tensor] = None, epilogue: int = EPI_NONE,
            aux: Optional[torch.Tensor] = None, aux_out: Optional[torch.Tensor] = None,
            out: Optional[torch.Tensor] = None, out_dtype=torch.bfloat16, alpha: float = 1.0,
            splitk: int = 1, accumulate: bool = False) -> torch.Tensor:
    """out[m, n] = epi(alpha * sum_k a[m, k] * w[n, k] + bias[n]).  a fp32|bf16, w bf16, fp32 accumulate."""
    _mat(a, "a"); _mat(w, "w")
    M, K = a.shape
    N, K2 = w.shape
    if K != K2 or w.dtype != torch.bfloat16:
        raise capi.UencError(f"gemm_nt: a {tuple(a.shape)} vs w {tuple(w.shape)} / {w.dtype}")
    if out is None:
        out = torch.empty((M, N), dtype=out_dtype, device=a.device)
        if splitk > 1 or accumulate:
            out.zero_()
    _mat(out, "out")
    assert out.shape == (M, N)
    if bias is not None:
        assert bias.dtype == torch.float32 and bias.numel() == N and bias.is_contiguous()
    if aux is not None:
        _mat(aux, "aux"); assert aux.shape == (M, N)
        assert aux.dtype == (torch.float32 if epilogue == EPI_RESIDUAL else torch.bfloat16)
    if aux_out is not None:
        _mat(aux_out, "aux_out"); assert aux_out.shape == (M, N) and aux_out.dtype == torch.bfloat16
    check(lib.uenc_gemm_nt(a.data_ptr(), dt(a), a.stride(0), w.data_ptr(), w.stride(0), out.data_ptr(), dt(out),
                           out.stride(0), M, N, K, ptr(bias), epilogue, ptr(aux), aux.stride(0) if aux is not None else 0,
                           ptr(aux_out), aux_out.stride(0) if aux_out is not None else 0, float(alpha), int(splitk),
                           int(accumulate), stream_ptr()), "gemm_nt")
    return out


def gemm_tn(dy: torch.Tensor, x: torch.Tensor, dw: torch.Tensor, db: Optional[torch.Tensor] = None, splitm: int = 0):
    """dw[n, k] += sum_m dy[m, n] * x[m, k];  db[n] += sum_m dy[m, n].  dy bf16, x fp32|bf16, dw/db fp32 (accumulated)."""
    _mat(dy, "dy"); _mat(x, "x"); _mat(dw, "dw")
    M, N = dy.shape
    M2, K = x.shape
    if M != M2 or dy.dtype != torch.bfloat16 or dw.dtype != torch.float32 or dw.shape != (N, K):
        raise capi.UencError(f"gemm_tn: dy {tuple(dy.shape)} x {tuple(x.shape)} dw {tuple(dw.shape)}")
    if db is not None:
        assert db.dtype == torch.float32 and db.numel() == N and db.is_contiguous()
    check(lib.uenc_gemm_tn(dy.data_ptr(), dy.stride(0), x.data_ptr(), dt(x), x.stride(0), dw.data_ptr(), dw.stride(0),
                           ptr(db), M, N, K, int(splitm), stream_ptr()), "gemm_tn")


def layernorm_fwd(x: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, *, res: Optional[torch.Tensor] = None,
                  out_dtype=torch.bfloat16, want_h: bool = False, want_stats: bool = True, eps: float = 1e-5):
    """y = LN(x + res).  Returns (y, h, stats): h = x + res in fp32 if want_h, stats = (M, 2) (mean, rstd)."""
    C = x.shape[-1]
    assert x.is_cuda and x.is_contiguous() and gamma.dtype == torch.float32 and beta.dtype == torch.float32
    M = x.numel() // C
    y = torch.empty(x.shape, dtype=out_dtype, device=x.device)
    h = torch.empty(x.shape, dtype=torch.float32, device=x.device) if want_h else None
    stats = torch.empty((M, 2), dtype=torch.float32, device=x.device) if want_stats else None
    if res is not None:
        assert res.shape == x.shape and res.is_contiguous()
    check(lib.uenc_layernorm_fwd(x.data_ptr(), dt(x), ptr(res), dt(res) if res is not None else 0, ptr(h),
                                 gamma.data_ptr(), beta.data_ptr(), y.data_ptr(), dt(y), ptr(stats), M, C, float(eps),
                                 stream_ptr()), "layernorm_fwd")
    return y, h, stats


def layernorm_bwd(dy: torch.Tensor, h: torch.Tensor, stats: torch.Tensor, gamma: torch.Tensor, *,
                  dres: Optional[torch.Tensor] = None, dgamma: Optional[torch.Tensor] = None,
                  dbeta: Optional[torch.Tensor] = None, dx_dtype=torch.float32) -> torch.Tensor:
    """dx = LN'(dy) [+ dres]; dgamma / dbeta are accumulated in place (fp32)."""
    C = h.shape[-1]
    M = h.numel() // C
    assert dy.is_contiguous() and h.is_contiguous() and dy.shape == h.shape
    dx = torch.empty(h.shape, dtype=dx_dtype, device=h.device)
    if dres is not None:
        assert dres.dtype == torch.float32 and dres.is_contiguous() and dres.shape == h.shape
    check(lib.uenc_layernorm_bwd(dy.data_ptr(), dt(dy), h.data_ptr(), dt(h), stats.data_ptr(), gamma.data_ptr(),
                                 ptr(dres), dx.data_ptr(), dt(dx), ptr(dgamma), ptr(dbeta), M, C, stream_ptr()),
          "layernorm_bwd")
    return dx
