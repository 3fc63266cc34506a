"""Thin launch wrappers: torch tensors in, C-ABI calls out.  No autograd here (see `ops.py`).

Every wrapper validates what the kernel's grid assumes (device, dtype, inner stride 1) before
handing raw pointers to the library; the library re-checks alignment and sizes and refuses
(-1) rather than launching on a bad shape.
"""
import os
from typing import Optional

import torch

from . import capi
from .capi import BF16, F32, check, dt, lib, ptr, stream_ptr

EPI_NONE, EPI_GELU, EPI_RELU, EPI_RESIDUAL, EPI_MUL_DGELU, EPI_MUL_DRELU = (
    capi.EPI_NONE, capi.EPI_GELU, capi.EPI_RELU, capi.EPI_RESIDUAL, capi.EPI_MUL_DGELU, capi.EPI_MUL_DRELU)


# fp32 "exact" arithmetic mode (csrc/exact.hip): every activation that is bf16 between kernels in the
# product becomes fp32 and every contraction runs on fp32 operands.  A verification mode: UENC_EXACT=1 or ops.set_exact(True).
EXACT = os.environ.get("UENC_EXACT", "0") == "1"


def adt():
    """dtype of activations handed from kernel to kernel: bf16, or fp32 in exact mode."""
    return torch.float32 if EXACT else torch.bfloat16


def _odt(dtype):
    return torch.float32 if (EXACT and dtype == torch.bfloat16) else dtype


def _mat(t: torch.Tensor, name: str):
    if not t.is_cuda:
        raise capi.UencError(f"{name} must live on the GPU")
    if t.dim() != 2 or t.stride(1) != 1:
        raise capi.UencError(f"{name} must be 2-D with unit inner stride, got {tuple(t.shape)} / {t.stride()}")
    return t


def dropout_bf16(x16: torch.Tensor, seed: int, p: float, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Inverted dropout of a contiguous bf16 tensor with the index-hash keep mask (uenc_dropout_bf16); out=x16 for in place."""
    assert x16.dtype == torch.bfloat16 and x16.is_contiguous() and x16.is_cuda and x16.numel() % 8 == 0
    if out is None:
        out = torch.empty_like(x16)
    assert out.dtype == torch.bfloat16 and out.is_contiguous() and out.numel() == x16.numel()
    check(lib.uenc_dropout_bf16(x16.data_ptr(), out.data_ptr(), x16.numel(), int(seed) & 0xFFFFFFFF, float(p), stream_ptr()), "dropout_bf16")
    return out


def dropout_keep_reference(shape, p: float, seed: int) -> torch.Tensor:
    """The keep-mask uenc_dropout_bf16 derives for a tensor of `shape` (row-major index), restated on the host.  Test helper."""
    from .attention import keep_mask_reference
    n = 1
    for d in shape:
        n *= int(d)
    return keep_mask_reference(1, 1, 1, n, p, int(seed) & 0xFFFFFFFF).view(tuple(shape))


def cast_bf16(src: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """fp32 -> bf16 copy (weights).  numel must be a multiple of 8, else falls to the padded path."""
    assert src.dtype == torch.float32 and src.is_contiguous() and src.is_cuda
    if EXACT:                                   # the "bf16 copy" of an fp32 tensor is the tensor itself
        return src if out is None else out.copy_(src)
    if out is None:
        out = torch.empty(src.shape, dtype=torch.bfloat16, device=src.device)
    n = src.numel()
    if n % 8 == 0 and src.data_ptr() % 16 == 0:
        check(lib.uenc_cast_f32_bf16(src.data_ptr(), out.data_ptr(), n, stream_ptr()), "cast_f32_bf16")
    else:  # tiny odd-sized vectors (biases of odd length): treated as a 1 x n transpose
        check(lib.uenc_cast_transpose_f32_bf16(src.data_ptr(), out.data_ptr(), 1, n, stream_ptr()), "cast_f32_bf16")
    return out


def cast_transpose_bf16(src: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """(R, C) fp32 -> (C, R) bf16."""
    assert src.dtype == torch.float32 and src.dim() == 2 and src.is_contiguous() and src.is_cuda
    R, C = src.shape
    if EXACT:                                   # data movement only
        return src.t().contiguous() if out is None else out.copy_(src.t())
    if out is None:
        out = torch.empty((C, R), dtype=torch.bfloat16, device=src.device)
    assert out.dtype == torch.bfloat16 and tuple(out.shape) == (C, R) and out.is_contiguous()
    check(lib.uenc_cast_transpose_f32_bf16(src.data_ptr(), out.data_ptr(), R, C, stream_ptr()), "cast_transpose")
    return out


def gemm_nt(a: torch.Tensor, w: torch.Tensor, *, bias: Optional[torch.Tensor] = None, epilogue: int = EPI_NONE,
            aux: Optional[torch.Tensor] = None, aux_out: Optional[torch.Tensor] = None,
            out: Optional[torch.Tensor] = None, out_dtype=torch.bfloat16, alpha: float = 1.0,
            splitk: int = 1, accumulate: bool = False, sample_scale: Optional[torch.Tensor] = None, rows_per_sample: int = 0) -> torch.Tensor:
    """out[m, n] = epi(alpha * (sum_k a[m, k] * w[n, k] + bias[n])).  a fp32|bf16, w bf16, fp32 accumulate.
    sample_scale (device fp32, one entry per `rows_per_sample` rows): alpha is multiplied by the row's sample scale (DropPath)."""
    _mat(a, "a"); _mat(w, "w")
    M, K = a.shape
    N, K2 = w.shape
    if EXACT:
        if sample_scale is not None:
            raise capi.UencError("gemm_nt: the fp32 verification kernels take no per-sample scale")
        return _gemm_nt_exact(a, w, bias, epilogue, aux, aux_out, out, alpha, accumulate)
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
    if sample_scale is not None:
        assert sample_scale.dtype == torch.float32 and sample_scale.is_cuda and sample_scale.is_contiguous()
        assert rows_per_sample > 0 and sample_scale.numel() * rows_per_sample >= M and splitk == 1 and not accumulate
        check(lib.uenc_gemm_nt_scaled(a.data_ptr(), dt(a), a.stride(0), w.data_ptr(), w.stride(0), out.data_ptr(), dt(out),
                                      out.stride(0), M, N, K, ptr(bias), epilogue, ptr(aux), aux.stride(0) if aux is not None else 0,
                                      ptr(aux_out), aux_out.stride(0) if aux_out is not None else 0, float(alpha), sample_scale.data_ptr(),
                                      int(rows_per_sample), stream_ptr()), "gemm_nt_scaled")
        return out
    check(lib.uenc_gemm_nt(a.data_ptr(), dt(a), a.stride(0), w.data_ptr(), w.stride(0), out.data_ptr(), dt(out),
                           out.stride(0), M, N, K, ptr(bias), epilogue, ptr(aux), aux.stride(0) if aux is not None else 0,
                           ptr(aux_out), aux_out.stride(0) if aux_out is not None else 0, float(alpha), int(splitk),
                           int(accumulate), stream_ptr()), "gemm_nt")
    return out


def _gemm_nt_exact(a, w, bias, epilogue, aux, aux_out, out, alpha, accumulate) -> torch.Tensor:
    """fp32 operands, fp32 result (uenc_gemm_nt_f32); same epilogues as the bf16 kernels, aux / aux_out fp32."""
    M, K = a.shape
    N, K2 = w.shape
    f32 = torch.float32
    if K != K2 or a.dtype != f32 or w.dtype != f32:
        raise capi.UencError(f"gemm_nt (exact): a {tuple(a.shape)} {a.dtype} vs w {tuple(w.shape)} {w.dtype}")
    if out is None:
        out = torch.zeros((M, N), dtype=f32, device=a.device) if accumulate else torch.empty((M, N), dtype=f32, device=a.device)
    _mat(out, "out")
    assert out.shape == (M, N) and out.dtype == f32
    for t, name in ((aux, "aux"), (aux_out, "aux_out")):
        if t is not None:
            _mat(t, name); assert t.shape == (M, N) and t.dtype == f32
    if bias is not None:
        assert bias.dtype == f32 and bias.numel() == N and bias.is_contiguous()
    check(lib.uenc_gemm_nt_f32(a.data_ptr(), a.stride(0), w.data_ptr(), w.stride(0), out.data_ptr(), out.stride(0), M, N, K, ptr(bias),
                               epilogue, ptr(aux), aux.stride(0) if aux is not None else 0, ptr(aux_out),
                               aux_out.stride(0) if aux_out is not None else 0, float(alpha), int(accumulate), stream_ptr()), "gemm_nt_f32")
    return out


def gemm_nt_splitk(a: torch.Tensor, w: torch.Tensor, splitk: int, *, alpha: float = 1.0) -> torch.Tensor:
    """sum_k a[m, k] * w[n, k] in fp32 for a long contraction with few output tiles: `splitk` k-ranges computed by separate
    workgroups, each STORING its partial tile (no atomics), then summed.  a, w bf16 with unit inner stride."""
    _mat(a, "a"); _mat(w, "w")
    M, K = a.shape
    N, K2 = w.shape
    if EXACT:
        return _gemm_nt_exact(a, w, None, EPI_NONE, None, None, None, alpha, False)
    if K != K2 or w.dtype != torch.bfloat16:
        raise capi.UencError(f"gemm_nt_splitk: a {tuple(a.shape)} vs w {tuple(w.shape)} / {w.dtype}")
    s = lib.uenc_gemm_nt_splits(K, int(splitk))
    part = torch.empty((s, M, N), dtype=torch.float32, device=a.device)
    check(lib.uenc_gemm_nt_partials(a.data_ptr(), dt(a), a.stride(0), w.data_ptr(), w.stride(0), part.data_ptr(), N, M * N, M, N, K,
                                    float(alpha), s, stream_ptr()), "gemm_nt_partials")
    return part[0] if s == 1 else part.sum(0)


def gemm_nt_batched(a: torch.Tensor, w: torch.Tensor, out: torch.Tensor, *, alpha: float = 1.0, splitk: int = 1, accumulate: bool = False):
    """out[b] (+)= a[b] @ w[b]^T for b in range(B): a (B, M, K) fp32|bf16, w (B, N, K) bf16, out (B, M, N) fp32|bf16, inner strides 1.
    One launch; with splitk > 1 / accumulate, out must be fp32 (zeroed by the caller for splitk > 1)."""
    B, M, K = a.shape
    _, N, _ = w.shape
    if EXACT:
        for b in range(B):
            _gemm_nt_exact(a[b], w[b], None, EPI_NONE, None, None, out[b], alpha, accumulate or splitk > 1)
        return out
    assert a.is_cuda and w.dtype == torch.bfloat16 and w.shape == (B, N, K) and out.shape == (B, M, N)
    assert a.stride(2) == 1 and w.stride(2) == 1 and out.stride(2) == 1
    check(lib.uenc_gemm_nt_batched(a.data_ptr(), dt(a), a.stride(1), a.stride(0), w.data_ptr(), w.stride(1), w.stride(0), out.data_ptr(), dt(out),
                                   out.stride(1), out.stride(0), B, M, N, K, float(alpha), int(splitk), int(accumulate), stream_ptr()),
          "gemm_nt_batched")
    return out


def gemm_tn(dy: torch.Tensor, x: torch.Tensor, dw: torch.Tensor, db: Optional[torch.Tensor] = None, splitm: int = 0, alpha: float = 1.0):
    """dw[n, k] += alpha * sum_m dy[m, n] * x[m, k];  db[n] += alpha * sum_m dy[m, n].  dy bf16, x fp32|bf16, dw/db fp32 (accumulated)."""
    _mat(dy, "dy"); _mat(x, "x"); _mat(dw, "dw")
    M, N = dy.shape
    M2, K = x.shape
    if M != M2 or dw.dtype != torch.float32 or dw.shape != (N, K):
        raise capi.UencError(f"gemm_tn: dy {tuple(dy.shape)} x {tuple(x.shape)} dw {tuple(dw.shape)}")
    if db is not None:
        assert db.dtype == torch.float32 and db.numel() == N and db.is_contiguous()
    if EXACT:
        assert dy.dtype == torch.float32 and x.dtype == torch.float32 and alpha == 1.0
        check(lib.uenc_gemm_tn_f32(dy.data_ptr(), dy.stride(0), x.data_ptr(), x.stride(0), dw.data_ptr(), dw.stride(0), ptr(db), M, N, K,
                                   stream_ptr()), "gemm_tn_f32")
        return
    if alpha != 1.0:
        check(lib.uenc_gemm_tn_scaled(dy.data_ptr(), dt(dy), dy.stride(0), x.data_ptr(), dt(x), x.stride(0), dw.data_ptr(), dw.stride(0),
                                      ptr(db), M, N, K, int(splitm), float(alpha), stream_ptr()), "gemm_tn_scaled")
        return
    check(lib.uenc_gemm_tn(dy.data_ptr(), dt(dy), dy.stride(0), x.data_ptr(), dt(x), x.stride(0), dw.data_ptr(), dw.stride(0),
                           ptr(db), M, N, K, int(splitm), stream_ptr()), "gemm_tn")


def gemm_nt_ln(a: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor], residual: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor,
               eps: float = 1e-5, want_y32: bool = True, want_y16: bool = True):
    """LayerNorm(a @ w^T + bias + residual) with the LayerNorm inside the GEMM's epilogue (one pass over the row instead of GEMM -> h ->
    LayerNorm kernel).  a (M, K) bf16, w (N, K) bf16, residual (M, N) fp32, N <= 256.
    -> (h fp32 pre-norm sum, y fp32 | None, y16 bf16 | None, stats (M, 2)) or None when the shape is not one the fused kernels take (the
    caller then runs gemm_nt + layernorm_fwd); never in the fp32 verification mode."""
    M, Kd = a.shape
    N = w.shape[0]
    if (EXACT or a.dtype != torch.bfloat16 or N > 256 or N % 8 or Kd % 64 or os.environ.get("UENC_GEMM_LN", "1") == "0"
            or not (a.stride(1) == 1 and w.stride(1) == 1 and residual.is_contiguous() and residual.dtype == torch.float32)):
        return None
    h = torch.empty((M, N), dtype=torch.float32, device=a.device)
    y32 = torch.empty((M, N), dtype=torch.float32, device=a.device) if want_y32 else None
    y16 = torch.empty((M, N), dtype=torch.bfloat16, device=a.device) if want_y16 else None
    stats = torch.empty((M, 2), dtype=torch.float32, device=a.device)
    rc = lib.uenc_gemm_nt_ln(a.data_ptr(), dt(a), a.stride(0), w.data_ptr(), w.stride(0), h.data_ptr(), N, M, N, Kd, ptr(bias), residual.data_ptr(), N,
                             gamma.data_ptr(), beta.data_ptr(), float(eps), ptr(y32), ptr(y16), stats.data_ptr(), stream_ptr())
    if rc == -1:
        return None
    check(rc, "gemm_nt_ln")
    return h, y32, y16, stats


def layernorm_fwd(x: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, *, res: Optional[torch.Tensor] = None,
                  out_dtype=torch.bfloat16, want_h: bool = False, want_stats: bool = True, eps: float = 1e-5,
                  twin: Optional[list] = None):
    """y = LN(x + res).  Returns (y, h, stats): h = x + res in fp32 if want_h, stats = (M, 2) (mean, rstd).
    twin: a list that receives the bf16 copy of an fp32 y, written in the same pass."""
    C = x.shape[-1]
    assert x.is_cuda and x.is_contiguous() and gamma.dtype == torch.float32 and beta.dtype == torch.float32
    M = x.numel() // C
    out_dtype = _odt(out_dtype)
    if EXACT and twin is not None:
        twin_exact, twin = twin, None           # the operand copy of an fp32 y is y itself
    else:
        twin_exact = None
    y = torch.empty(x.shape, dtype=out_dtype, device=x.device)
    h = torch.empty(x.shape, dtype=torch.float32, device=x.device) if want_h else None
    stats = torch.empty((M, 2), dtype=torch.float32, device=x.device) if want_stats else None
    if res is not None:
        assert res.shape == x.shape and res.is_contiguous()
    y16 = None
    if twin is not None and out_dtype == torch.float32:
        y16 = torch.empty(x.shape, dtype=torch.bfloat16, device=x.device)
        twin.append(y16)
    check(lib.uenc_layernorm_fwd(x.data_ptr(), dt(x), ptr(res), dt(res) if res is not None else 0, ptr(h),
                                 gamma.data_ptr(), beta.data_ptr(), y.data_ptr(), dt(y), ptr(stats), M, C, float(eps),
                                 ptr(y16), stream_ptr()), "layernorm_fwd")
    if twin_exact is not None and out_dtype == torch.float32:
        twin_exact.append(y)
    return y, h, stats


def layernorm_bwd(dy: torch.Tensor, h: torch.Tensor, stats: torch.Tensor, gamma: torch.Tensor, *,
                  dres: Optional[torch.Tensor] = None, dgamma: Optional[torch.Tensor] = None,
                  dbeta: Optional[torch.Tensor] = None, dx_dtype=torch.float32, twin: Optional[list] = None,
                  defer: Optional["SmallReductions"] = None) -> torch.Tensor:
    """dx = LN'(dy) [+ dres]; dgamma / dbeta are accumulated in place (fp32).  twin: a list that receives the bf16 copy of an
    fp32 dx, written in the same pass.  defer: a SmallReductions queue -- the dgamma / dbeta block partials are parked in a private
    buffer and summed by the queue's next flush (one launch for all parked passes) instead of by a launch of their own."""
    C = h.shape[-1]
    M = h.numel() // C
    assert dy.is_contiguous() and h.is_contiguous() and dy.shape == h.shape
    dx_dtype = _odt(dx_dtype)
    twin_exact = None
    if EXACT and twin is not None:
        twin_exact, twin = twin, None
    dx = torch.empty(h.shape, dtype=dx_dtype, device=h.device)
    if dres is not None:
        assert dres.dtype == torch.float32 and dres.is_contiguous() and dres.shape == h.shape
    dx16 = None
    if twin is not None and dx_dtype == torch.float32:
        dx16 = torch.empty(h.shape, dtype=torch.bfloat16, device=h.device)
        twin.append(dx16)
    part, deferred = _ln_part(M, C, dgamma, dbeta, h.device, defer)
    check(lib.uenc_layernorm_bwd(dy.data_ptr(), dt(dy), h.data_ptr(), dt(h), stats.data_ptr(), gamma.data_ptr(),
                                 ptr(dres), dx.data_ptr(), dt(dx), ptr(dgamma), ptr(dbeta), M, C, ptr(dx16), ptr(part), int(deferred), stream_ptr()),
          "layernorm_bwd")
    if twin_exact is not None and dx_dtype == torch.float32:
        twin_exact.append(dx)
    return dx


def postproc_semantic(mask_logits: torch.Tensor, class_prob: torch.Tensor, padded_size, out_size) -> torch.Tensor:
    """mask_logits (Q, h, w) fp32, class_prob (Q, C) fp32 -> (C, Ho, Wo) fp32 = sum_q p[q, c] * sigmoid(upsample(m_q)), upsampled
    to `padded_size` and cropped to `out_size` without materialising the (Q, H, W) masks."""
    Q, hl, wl = mask_logits.shape
    C = class_prob.shape[1]
    assert mask_logits.dtype == torch.float32 and mask_logits.is_contiguous() and class_prob.shape[0] == Q
    Cp = -(-C // 32) * 32
    P = torch.zeros((Q, Cp), dtype=torch.float32, device=mask_logits.device)
    P[:, :C] = class_prob
    sem = torch.empty((C, out_size[0], out_size[1]), dtype=torch.float32, device=mask_logits.device)
    check(lib.uenc_postproc_semantic(mask_logits.data_ptr(), P.data_ptr(), sem.data_ptr(), Q, C, Cp, hl, wl, padded_size[0], padded_size[1],
                                     out_size[0], out_size[1], stream_ptr()), "postproc_semantic")
    return sem


def postproc_panoptic_stats(mask_logits: torch.Tensor, score: torch.Tensor, padded_size, out_size):
    """-> ids (Ho, Wo) int32 (argmax over queries with score > 0 of score * sigmoid(upsampled mask)), counts (3, Q) int32:
    pixels won, pixels with sigmoid >= 0.5, pixels with both."""
    Q, hl, wl = mask_logits.shape
    assert mask_logits.dtype == torch.float32 and mask_logits.is_contiguous() and score.dtype == torch.float32 and score.numel() == Q
    ids = torch.empty(tuple(out_size), dtype=torch.int32, device=mask_logits.device)
    counts = torch.zeros((3, Q), dtype=torch.int32, device=mask_logits.device)
    check(lib.uenc_postproc_panoptic_stats(mask_logits.data_ptr(), score.contiguous().data_ptr(), ids.data_ptr(), counts.data_ptr(), Q, hl, wl,
                                           padded_size[0], padded_size[1], out_size[0], out_size[1], stream_ptr()), "postproc_panoptic_stats")
    return ids, counts


def postproc_panoptic_label(mask_logits: torch.Tensor, ids: torch.Tensor, segid: torch.Tensor, padded_size) -> torch.Tensor:
    Q, hl, wl = mask_logits.shape
    assert ids.dtype == torch.int32 and segid.dtype == torch.int32 and segid.numel() == Q and ids.is_contiguous()
    seg = torch.empty_like(ids)
    check(lib.uenc_postproc_panoptic_label(mask_logits.data_ptr(), ids.data_ptr(), segid.contiguous().data_ptr(), seg.data_ptr(), Q, hl, wl,
                                           padded_size[0], padded_size[1], ids.shape[0], ids.shape[1], stream_ptr()), "postproc_panoptic_label")
    return seg


def im2col3x3_s2(x16: torch.Tensor) -> torch.Tensor:
    """x (B, H, W, C) bf16 channels-last -> patch matrix (B * ceil(H/2) * ceil(W/2), 9C) bf16 of the 3x3 stride-2 pad-1 convolution."""
    B, H, W, C = x16.shape
    if EXACT and x16.dtype == torch.float32:     # pure data movement: the fp32 map gathered as a bf16 map with twice the channels
        return im2col3x3_s2(x16.contiguous().view(torch.bfloat16)).view(torch.float32)
    assert x16.dtype == torch.bfloat16 and x16.is_contiguous() and C % 8 == 0
    col = torch.empty((B * ((H + 1) // 2) * ((W + 1) // 2), 9 * C), dtype=torch.bfloat16, device=x16.device)
    check(lib.uenc_im2col3x3_s2(x16.data_ptr(), col.data_ptr(), B, H, W, C, stream_ptr()), "im2col3x3_s2")
    return col


def col2im3x3_s2(dcol: torch.Tensor, B: int, H: int, W: int, C: int) -> torch.Tensor:
    """adjoint of im2col3x3_s2: (rows, 9C) bf16 -> (B, H, W, C) fp32."""
    assert dcol.dtype == torch.bfloat16 and dcol.is_contiguous() and dcol.shape[1] == 9 * C
    dx = torch.empty((B, H, W, C), dtype=torch.float32, device=dcol.device)
    check(lib.uenc_col2im3x3_s2(dcol.data_ptr(), dx.data_ptr(), B, H, W, C, stream_ptr()), "col2im3x3_s2")
    return dx


def na2d_fwd(qkv: torch.Tensor, rpb: Optional[torch.Tensor], nH: int, ks: int, dilation: int, scale: float, need_lse: bool = True):
    """Neighbourhood attention on qkv (B, H, W, 3C) bf16 (C = nH * 32) -> out (B, H, W, C) bf16, lse (B, nH, H, W) fp32."""
    B, H, W, C3 = qkv.shape
    C = C3 // 3
    if EXACT:
        raise NotImplementedError("the fp32 exact mode covers the Swin path; neighbourhood attention has no fp32 kernel")
    assert qkv.dtype == torch.bfloat16 and qkv.is_contiguous() and C == nH * 32, "na2d: head_dim must be 32"
    if rpb is not None:
        assert rpb.dtype == torch.float32 and rpb.is_contiguous() and tuple(rpb.shape) == (nH, 2 * ks - 1, 2 * ks - 1)
    out = torch.empty((B, H, W, C), dtype=torch.bfloat16, device=qkv.device)
    lse = torch.empty((B, nH, H, W), dtype=torch.float32, device=qkv.device) if need_lse else None
    check(lib.uenc_na2d_fwd(qkv.data_ptr(), ptr(rpb), out.data_ptr(), ptr(lse), B, H, W, nH, ks, dilation, float(scale), stream_ptr()),
          "na2d_fwd")
    return out, lse


def na2d_bwd(qkv, rpb, out, dout, lse, nH: int, ks: int, dilation: int, scale: float, drpb: Optional[torch.Tensor]):
    """-> dqkv (B, H, W, 3C) bf16; drpb (nH, 2ks-1, 2ks-1) fp32 is accumulated in place when given."""
    B, H, W, C3 = qkv.shape
    assert dout.dtype == torch.bfloat16 and dout.is_contiguous() and out.is_contiguous() and dout.shape == out.shape
    dqkv = torch.empty_like(qkv)
    ws = _scratch("na2d_delta", B * nH * H * W * 4, qkv.device)
    check(lib.uenc_na2d_bwd(qkv.data_ptr(), ptr(rpb), out.data_ptr(), dout.data_ptr(), lse.data_ptr(), dqkv.data_ptr(), ptr(drpb),
                            ws.data_ptr(), B, H, W, nH, ks, dilation, float(scale), stream_ptr()), "na2d_bwd")
    return dqkv


def patch_merge_ln_fwd(x: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, eps: float = 1e-5):
    """x (B, H, W, C) fp32 -> (y (B, ceil(H/2) * ceil(W/2), 4C) bf16, stats): PatchMerging's gather + LayerNorm in one pass."""
    B, H, W, C = x.shape
    assert x.dtype == torch.float32 and x.is_contiguous() and gamma.numel() == 4 * C
    L2 = ((H + 1) // 2) * ((W + 1) // 2)
    y = torch.empty((B, L2, 4 * C), dtype=torch.bfloat16, device=x.device)
    stats = torch.empty((B * L2, 2), dtype=torch.float32, device=x.device)
    check(lib.uenc_patch_merge_ln_fwd(x.data_ptr(), gamma.data_ptr(), beta.data_ptr(), y.data_ptr(), stats.data_ptr(), B, H, W, C, float(eps),
                                      stream_ptr()), "patch_merge_ln_fwd")
    return y, stats


def patch_merge_ln_bwd(dy: torch.Tensor, x: torch.Tensor, stats: torch.Tensor, gamma: torch.Tensor, dgamma=None, dbeta=None,
                       defer: Optional["SmallReductions"] = None) -> torch.Tensor:
    B, H, W, C = x.shape
    assert dy.is_contiguous() and dy.shape[-1] == 4 * C
    dx = torch.empty_like(x)
    M = B * ((H + 1) // 2) * ((W + 1) // 2)
    part, deferred = _ln_part(M, 4 * C, dgamma, dbeta, x.device, defer)
    check(lib.uenc_patch_merge_ln_bwd(dy.data_ptr(), dt(dy), x.data_ptr(), stats.data_ptr(), gamma.data_ptr(), dx.data_ptr(), ptr(dgamma),
                                      ptr(dbeta), ptr(part), B, H, W, C, int(deferred), stream_ptr()), "patch_merge_ln_bwd")
    return dx


class SmallReductions:
    """Parameter-gradient reductions that nothing in the backward pass waits for, parked and launched together.

    A LayerNorm backward leaves [blocks][2][C] partial sums of dgamma / dbeta, a window-attention backward dense dS partials of its
    relative-position table; summing them is ~2 us of work in a ~10 / ~19 us launch, ~70 + 24 times per step.  Only the optimiser (and the
    gradient all-reduce) reads the results, so the partials stay in private buffers and ONE grouped launch per kind sums them when the
    queue is flushed (ops.WgradQueue.flush: with the deferred weight-gradient groups, and at the end of every backward pass)."""
    _LN = [("part", "<u8"), ("dgamma", "<u8"), ("dbeta", "<u8"), ("nblk", "<i4"), ("C", "<i4"), ("begin", "<i4"), ("pad", "<i4")]
    _DT = [("wsd", "<u8"), ("dtab", "<u8"), ("G", "<i4"), ("nH", "<i4"), ("ws", "<i4"), ("ntiles", "<i4"), ("begin", "<i4"), ("pad", "<i4")]

    CHUNK = 64 << 20          # floats' worth of bytes per arena chunk

    def __init__(self):
        self.ln, self.dt, self.keep = [], [], []
        self._chunks, self._cur, self._off = [], 0, 0        # bump-allocated arena for the parked partials, reused every step

    def alloc(self, nfloats: int, device) -> torch.Tensor:
        """A private fp32 buffer that stays untouched until the next flush.  Bump-allocated from chunks that persist across steps (the
        ~1 GB of partials a backward pass parks would otherwise go through the caching allocator ~140 times per step)."""
        nbytes = (nfloats * 4 + 255) // 256 * 256
        while True:
            if self._cur < len(self._chunks):
                c = self._chunks[self._cur]
                if c.device == device and self._off + nbytes <= c.numel():
                    out = c[self._off:self._off + nfloats * 4].view(torch.float32)
                    self._off += nbytes
                    return out
                self._cur += 1
                self._off = 0
                continue
            self._chunks.append(torch.empty(max(self.CHUNK, nbytes), dtype=torch.uint8, device=device))

    def __bool__(self):
        return bool(self.ln or self.dt)

    def add_ln(self, part, dgamma, dbeta, nblk: int, C: int):
        self.ln.append((part.data_ptr(), dgamma.data_ptr(), dbeta.data_ptr(), nblk, C))
        self.keep.append((part, dgamma, dbeta))

    def add_dtable(self, wsd, dtab, G: int, nH: int, ws: int, ntiles: int):
        self.dt.append((wsd.data_ptr(), dtab.data_ptr(), G, nH, ws, ntiles))
        self.keep.append((wsd, dtab))

    def clear(self):
        self.ln, self.dt, self.keep = [], [], []
        self._cur, self._off = 0, 0           # stream order: whatever is parked next is written after the reductions launched above

    def flush(self):
        import numpy as np
        dev = self.keep[0][0].device if self.keep else None
        if self.ln:
            desc = np.zeros(len(self.ln), dtype=self._LN)
            begin = 0
            for i, (part, dg, db, nblk, C) in enumerate(self.ln):
                desc[i] = (part, dg, db, nblk, C, begin, 0)
                begin += -(-2 * C // 64)
            tab = torch.from_numpy(desc.view(np.uint8)).pin_memory().to(dev, non_blocking=True)
            check(lib.uenc_ln_param_grouped(tab.data_ptr(), len(self.ln), begin, stream_ptr()), "ln_param_grouped")
        if self.dt:
            desc = np.zeros(len(self.dt), dtype=self._DT)
            begin = 0
            for i, (wsd, dtab, G, nH, ws, nt) in enumerate(self.dt):
                desc[i] = (wsd, dtab, G, nH, ws, nt, begin, 0)
                begin += nH * nt
            tab = torch.from_numpy(desc.view(np.uint8)).pin_memory().to(dev, non_blocking=True)
            check(lib.uenc_window_attn_dtable_grouped(tab.data_ptr(), len(self.dt), begin, stream_ptr()), "window_attn_dtable_grouped")
        self.clear()


def _ln_part(M: int, C: int, dgamma, dbeta, device, defer):
    """Scratch for a LayerNorm backward's dgamma / dbeta block partials: the shared buffer (the pass reduces them itself), or -- deferred --
    a private one registered with the queue.  -> (buffer | None, deferred?)"""
    if dgamma is None:
        return None, False
    nblk = int(lib.uenc_layernorm_bwd_blocks(M, C)) if defer is not None else 0
    if nblk > 0:
        part = defer.alloc(nblk * 2 * C, device)
        defer.add_ln(part, dgamma, dbeta, nblk, C)
        return part, True
    return _scratch("ln_bwd", 2048 * 2 * C * 4, device), False


def relpos_expand(table: torch.Tensor, ws: int):
    """relative_position_bias_table ((2ws-1)^2, nH) fp32 -> dense (nH, NP, NP) in query-major and key-major order."""
    assert table.dtype == torch.float32 and table.is_contiguous() and table.shape[0] == (2 * ws - 1) ** 2
    if EXACT:                                   # the fp32 window-attention kernels index the table themselves
        return table, table
    nH = table.shape[1]
    NP = lib.uenc_window_attn_np(ws)
    bq = torch.empty((nH, NP, NP), dtype=torch.float32, device=table.device)
    bk = torch.empty_like(bq)
    check(lib.uenc_relpos_expand(table.data_ptr(), bq.data_ptr(), bk.data_ptr(), nH, ws, stream_ptr()), "relpos_expand")
    return bq, bk


def window_attn_fwd(qkv: torch.Tensor, qkv_bias16: torch.Tensor, bias_q: torch.Tensor, ws: int, shift: int,
                    scale: float, want_lse: bool = False):
    """qkv (B, H, W, 3C) bf16 -> attention output (B, H, W, C) bf16 (before proj).  head_dim is 32.
    want_lse: also return the softmax row statistics (B, H, W, nH) fp32 (max + log2 sum, log2 units) for window_attn_bwd(lse=...)."""
    B, H, W, C3 = qkv.shape
    C = C3 // 3
    if EXACT:                                   # qkv / bias fp32, bias_q = the raw relative-position table ((2ws-1)^2, nH)
        assert qkv.dtype == torch.float32 and qkv.is_contiguous() and qkv_bias16.dtype == torch.float32 and qkv_bias16.numel() == C3
        assert bias_q.dtype == torch.float32 and tuple(bias_q.shape) == ((2 * ws - 1) ** 2, C // 32) and bias_q.is_contiguous()
        out = torch.empty((B, H, W, C), dtype=torch.float32, device=qkv.device)
        check(lib.uenc_window_attn_f32_fwd(qkv.data_ptr(), qkv_bias16.data_ptr(), bias_q.data_ptr(), out.data_ptr(), B, H, W, C, C // 32, ws,
                                           shift, float(scale), stream_ptr()), "window_attn_f32_fwd")
        return (out, None) if want_lse else out
    assert qkv.dtype == torch.bfloat16 and qkv.is_contiguous() and qkv_bias16.dtype == torch.bfloat16
    assert qkv_bias16.numel() == C3 and C % 32 == 0 and bias_q.shape[0] == C // 32
    out = torch.empty((B, H, W, C), dtype=torch.bfloat16, device=qkv.device)
    lse = torch.empty((B, H, W, C // 32), dtype=torch.float32, device=qkv.device) if want_lse else None
    check(lib.uenc_window_attn_fwd(qkv.data_ptr(), qkv_bias16.data_ptr(), bias_q.data_ptr(), out.data_ptr(), lse.data_ptr() if want_lse else 0,
                                   B, H, W, C, C // 32, ws, shift, float(scale), stream_ptr()), "window_attn_fwd")
    return (out, lse) if want_lse else out


_SCRATCH = {}


def _scratch(tag: str, nbytes: int, device) -> torch.Tensor:
    """Kernel-internal scratch kept across calls: one buffer per (purpose, device), grown on demand.  Launches on one
    stream are ordered, so a buffer can be reused by the next call as soon as this one is enqueued."""
    key = (tag, str(device))
    buf = _SCRATCH.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty(max(nbytes, 256), dtype=torch.uint8, device=device)
        _SCRATCH[key] = buf
    return buf


def window_attn_bwd(qkv, qkv_bias16, bias_q, bias_k, o_saved, d_out, ws: int, shift: int, scale: float,
                    dtable: Optional[torch.Tensor] = None, dbias: Optional[torch.Tensor] = None, defer: Optional[SmallReductions] = None,
                    lse: Optional[torch.Tensor] = None):
    """-> dqkv (B,H,W,3C).  The two parameter gradients are ACCUMULATED by the kernels into `dtable` ((2ws-1)^2, nH) fp32 -- the
    relative-position table's .grad -- and `dbias` (3C) fp32 -- the share of qkv.bias.grad that flows through padding slots; when
    a buffer is not given (no training) the contribution goes to scratch."""
    B, H, W, C3 = qkv.shape
    C = C3 // 3
    nH = C // 32
    TT = (2 * ws - 1) ** 2
    if dtable is None:
        dtable = _scratch("wattn_dtab", TT * nH * 4, qkv.device).view(torch.float32)[:TT * nH].view(TT, nH)
    if dbias is None:
        dbias = _scratch("wattn_dpad", 3 * C * 4, qkv.device).view(torch.float32)[:3 * C]
    assert dtable.dtype == torch.float32 and dtable.is_contiguous() and tuple(dtable.shape) == (TT, nH)
    assert dbias.dtype == torch.float32 and dbias.is_contiguous() and dbias.numel() == 3 * C
    if EXACT:
        assert qkv.dtype == torch.float32 and d_out.dtype == torch.float32 and d_out.is_contiguous() and d_out.shape == (B, H, W, C)
        dqkv = torch.empty_like(qkv)
        check(lib.uenc_window_attn_f32_bwd(qkv.data_ptr(), qkv_bias16.data_ptr(), bias_q.data_ptr(), d_out.data_ptr(), dqkv.data_ptr(),
                                           dtable.data_ptr(), dbias.data_ptr(), B, H, W, C, nH, ws, shift, float(scale), stream_ptr()),
              "window_attn_f32_bwd")
        return dqkv
    assert d_out.dtype == torch.bfloat16 and d_out.is_contiguous() and d_out.shape == (B, H, W, C)
    assert o_saved.dtype == torch.bfloat16 and o_saved.is_contiguous()
    dqkv = torch.empty_like(qkv)
    nws = int(lib.uenc_window_attn_bwd_ws_floats(B, H, W, nH, ws))
    if defer is not None:                                                # the table gradient is reduced by the queue's grouped launch
        wsbuf = defer.alloc(nws, qkv.device)
        defer.add_dtable(wsbuf, dtable, int(lib.uenc_window_attn_bwd_groups(B, H, W, nH, ws)), nH, ws, int(lib.uenc_window_attn_np(ws)) // 16)
    else:
        wsbuf = _scratch("wattn_dS", nws * 4, qkv.device)                # dense dS partials (internal)
    if lse is not None:
        assert lse.dtype == torch.float32 and lse.is_contiguous() and tuple(lse.shape) == (B, H, W, nH)
    check(lib.uenc_window_attn_bwd(qkv.data_ptr(), qkv_bias16.data_ptr(), bias_q.data_ptr(), bias_k.data_ptr(),
                                   o_saved.data_ptr(), lse.data_ptr() if lse is not None else 0, d_out.data_ptr(), dqkv.data_ptr(), wsbuf.data_ptr(),
                                   dtable.data_ptr(), dbias.data_ptr(),
                                   B, H, W, C, nH, ws, shift, float(scale), int(defer is not None), stream_ptr()), "window_attn_bwd")
    return dqkv


def msdeform_tiled_eligible(value, shapes_host, Lq: int, L: int, P: int) -> bool:
    """Geometry of the deformable encoder (queries = the pixels of the L maps): the LDS-tiled forward applies."""
    if shapes_host is None or os.environ.get("UENC_MSDA_TILED", "1") == "0":
        return False
    B, S, M, D = value.shape
    return (D == 32 and value.dtype == torch.bfloat16 and L <= 4 and L * P <= 16 and Lq == S and B * M < 65536
            and sum(int(h) * int(w) for h, w in shapes_host) == S)


def msdeform_attn_fwd(value, shapes, level_start, loc, attn, out_dtype=torch.float32, shapes_host=None):
    """value (B,S,M,D) fp32|bf16, shapes (L,2) int64, level_start (L) int64, loc (B,Lq,M,L,P,2), attn (B,Lq,M,L,P)
    -> (B, Lq, M*D).  Same tensor contract as the reference's ms_deform_attn_forward.  shapes_host: optional [(H, W), ...] host copy
    of `shapes`; with it the encoder's geometry (Lq == S) takes the LDS-tiled kernel (same result)."""
    B, S, M, D = value.shape
    _, Lq, _, L, P, _ = loc.shape
    for t in (value, shapes, level_start, loc, attn):
        if not (t.is_cuda and t.is_contiguous()):
            raise capi.UencError("msdeform_attn: tensors must be contiguous CUDA tensors")
    assert shapes.dtype == torch.int64 and level_start.dtype == torch.int64
    assert loc.dtype == torch.float32 and attn.dtype == torch.float32 and attn.shape == (B, Lq, M, L, P)
    out = torch.empty((B, Lq, M * D), dtype=_odt(out_dtype), device=value.device)
    if msdeform_tiled_eligible(value, shapes_host, Lq, L, P):
        import ctypes
        flat = [int(v) for hw in shapes_host for v in hw]
        sh = (ctypes.c_int64 * len(flat))(*flat)
        check(lib.uenc_msdeform_attn_fwd_tiled(value.data_ptr(), dt(value), shapes.data_ptr(), level_start.data_ptr(), loc.data_ptr(),
                                               attn.data_ptr(), out.data_ptr(), dt(out), B, S, M, D, L, Lq, P, sh, stream_ptr()),
              "msdeform_attn_fwd_tiled")
        return out
    check(lib.uenc_msdeform_attn_fwd(value.data_ptr(), dt(value), shapes.data_ptr(), level_start.data_ptr(), loc.data_ptr(),
                                     attn.data_ptr(), out.data_ptr(), dt(out), B, S, M, D, L, Lq, P, stream_ptr()),
          "msdeform_attn_fwd")
    return out


def _msda_workspace(nbytes: int, device) -> torch.Tensor:
    """Scratch for the binned backward (bin counters + records)."""
    return _scratch("msda_bins", nbytes, device)


def msdeform_attn_bwd(value, shapes, level_start, loc, attn, grad_out, shapes_host=None):
    """-> grad_value (fp32, B,S,M,D), grad_loc, grad_attn (the reference's ms_deform_attn_backward contract).

    shapes_host: optional [(H, W), ...] host copy of `shapes`; enables the binned (on-chip summed) grad_value path."""
    B, S, M, D = value.shape
    _, Lq, _, L, P, _ = loc.shape
    assert grad_out.is_contiguous() and grad_out.shape == (B, Lq, M * D)
    gv = torch.zeros((B, S, M, D), dtype=torch.float32, device=value.device)
    gl = torch.empty_like(loc)
    ga = torch.empty_like(attn)
    sh, ws, ws_bytes = None, None, 0
    if shapes_host is not None:
        import ctypes
        flat = [int(v) for hw in shapes_host for v in hw]
        assert len(flat) == 2 * L
        sh = (ctypes.c_int64 * len(flat))(*flat)
        ws_bytes = int(lib.uenc_msdeform_attn_bwd_workspace_bytes(sh, B, M, D, L, Lq, P))
        if ws_bytes > 0:
            ws = _msda_workspace(ws_bytes, value.device)
    check(lib.uenc_msdeform_attn_bwd(value.data_ptr(), dt(value), shapes.data_ptr(), level_start.data_ptr(), loc.data_ptr(),
                                     attn.data_ptr(), grad_out.data_ptr(), dt(grad_out), gv.data_ptr(), gl.data_ptr(),
                                     ga.data_ptr(), B, S, M, D, L, Lq, P, sh, ws.data_ptr() if ws is not None else None,
                                     ws_bytes, stream_ptr()), "msdeform_attn_bwd")
    return gv, gl, ga


def _msda_host_plan(shapes_host, B, M, D, L, Lq, P):
    import ctypes
    flat = [int(v) for hw in shapes_host for v in hw]
    assert len(flat) == 2 * L
    sh = (ctypes.c_int64 * len(flat))(*flat)
    return sh, int(lib.uenc_msdeform_attn_bwd_workspace_bytes(sh, B, M, D, L, Lq, P))


def msdeform_fused_available(shapes_host, B, M, D, L, Lq, P) -> bool:
    """The fused forward / backward pair (locations and weights derived inside the kernels) needs D = 32, L * P <= 16 and the binned
    backward's plan; never in the fp32 verification mode (its reference path stays module by module)."""
    return (not EXACT) and D == 32 and L * P <= 16 and shapes_host is not None and _msda_host_plan(shapes_host, B, M, D, L, Lq, P)[1] > 0


def msdeform_attn_fused_fwd(value, shapes, level_start, offaw, ref, L: int, P: int, out_dtype=torch.bfloat16, shapes_host=None):
    """value (B, S, M, D), offaw (B * Lq, ld) fp32 = [M][L][P][2] offsets | [M][L * P] logits, ref (B | 1, Lq, L, 2) -> (B, Lq, M * D).
    shapes_host: optional host copy of `shapes`; with it the encoder's geometry (Lq == S, P == 4) takes the LDS-tiled kernel (same result)."""
    B, S, M, D = value.shape
    Lq = ref.shape[1]
    assert offaw.dtype == torch.float32 and offaw.stride(1) == 1 and offaw.shape[0] == B * Lq and ref.dtype == torch.float32 and ref.is_contiguous()
    out = torch.empty((B, Lq, M * D), dtype=_odt(out_dtype), device=value.device)
    if (P == 4 and offaw.stride(0) % 4 == 0 and offaw.data_ptr() % 16 == 0 and msdeform_tiled_eligible(value, shapes_host, Lq, L, P)):
        import ctypes
        flat = [int(v) for hw in shapes_host for v in hw]
        sh = (ctypes.c_int64 * len(flat))(*flat)
        check(lib.uenc_msdeform_attn_fused_fwd_tiled(value.data_ptr(), dt(value), shapes.data_ptr(), level_start.data_ptr(), offaw.data_ptr(), offaw.stride(0),
                                                     ref.data_ptr(), int(ref.shape[0] != 1), out.data_ptr(), dt(out), B, S, M, D, L, Lq, P, sh, stream_ptr()),
              "msdeform_attn_fused_fwd_tiled")
        return out
    check(lib.uenc_msdeform_attn_fused_fwd(value.data_ptr(), dt(value), shapes.data_ptr(), level_start.data_ptr(), offaw.data_ptr(), offaw.stride(0),
                                           ref.data_ptr(), int(ref.shape[0] != 1), out.data_ptr(), dt(out), B, S, M, D, L, Lq, P, stream_ptr()),
          "msdeform_attn_fused_fwd")
    return out


def msdeform_attn_fused_bwd(value, shapes, level_start, offaw, ref, L: int, P: int, grad_out, shapes_host):
    """-> grad_value (fp32, B, S, M, D), d(offaw) (B * Lq, 3 M L P) bf16."""
    B, S, M, D = value.shape
    Lq = ref.shape[1]
    assert grad_out.is_contiguous() and grad_out.shape == (B, Lq, M * D)
    sh, ws_bytes = _msda_host_plan(shapes_host, B, M, D, L, Lq, P)
    assert ws_bytes > 0
    ws = _msda_workspace(ws_bytes, value.device)
    gv = torch.zeros((B, S, M, D), dtype=torch.float32, device=value.device)
    ncol = 3 * M * L * P
    doffaw = torch.empty((B * Lq, ncol), dtype=torch.bfloat16, device=value.device)
    check(lib.uenc_msdeform_attn_fused_bwd(value.data_ptr(), dt(value), shapes.data_ptr(), level_start.data_ptr(), offaw.data_ptr(), offaw.stride(0),
                                           ref.data_ptr(), int(ref.shape[0] != 1), grad_out.data_ptr(), dt(grad_out), gv.data_ptr(), doffaw.data_ptr(), doffaw.stride(0),
                                           B, S, M, D, L, Lq, P, sh, ws.data_ptr(), ws_bytes, stream_ptr()), "msdeform_attn_fused_bwd")
    return gv, doffaw


def attn_mask(logits: torch.Tensor, size) -> torch.Tensor:
    """(B, Q, Hi, Wi) fp32 mask logits -> (B, Q, Ho*Wo) bool, True = blocked; fully blocked rows are cleared."""
    B, Q, Hi, Wi = logits.shape
    Ho, Wo = int(size[0]), int(size[1])
    assert logits.dtype == torch.float32 and logits.is_cuda and logits.is_contiguous()
    out = torch.empty((B, Q, Ho * Wo), dtype=torch.bool, device=logits.device)
    check(lib.uenc_attn_mask(logits.data_ptr(), out.data_ptr(), B * Q, Hi, Wi, Ho, Wo, stream_ptr()), "attn_mask")
    return out


def upsample_bilinear(x: torch.Tensor, size) -> torch.Tensor:
    """(N, C, Hi, Wi) fp32 -> (N, C, Ho, Wo), bilinear, align_corners=False (forward only; Wo % 4 == 0)."""
    N, C, Hi, Wi = x.shape
    Ho, Wo = int(size[0]), int(size[1])
    assert x.dtype == torch.float32 and x.is_cuda and Wo % 4 == 0
    x = x.contiguous()
    out = torch.empty((N, C, Ho, Wo), dtype=torch.float32, device=x.device)
    check(lib.uenc_upsample_bilinear(x.data_ptr(), out.data_ptr(), N * C, Hi, Wi, Ho, Wo, stream_ptr()), "upsample_bilinear")
    return out


# --------------------------------------------------------------------------------------------
# FPN branch on token matrices
# --------------------------------------------------------------------------------------------
def groupnorm_tokens_fwd(x, gamma, beta, G: int, eps: float, *, relu=False, add_src=None, add_hw=None, out_dtype=torch.float32):
    """x (B, HW, C) fp32|bf16 -> (y (B, HW, C) out_dtype, stats (B, G, 2)).  add_src: fp32 (B, Hs, Ws, C) merged in by
    bilinear resize to add_hw = (H, W)."""
    B, HW, C = x.shape
    assert x.is_cuda and x.is_contiguous() and gamma.dtype == torch.float32 and beta.dtype == torch.float32
    y = torch.empty((B, HW, C), dtype=_odt(out_dtype), device=x.device)
    stats = torch.empty((B, G, 2), dtype=torch.float32, device=x.device)
    scratch = _scratch("gn", int(lib.uenc_groupnorm_tokens_scratch_bytes(B, HW, C, G)), x.device)
    Hs = Ws = H = W = 0
    if add_src is not None:
        assert add_src.dtype == torch.float32 and add_src.is_contiguous() and add_src.shape[0] == B and add_src.shape[3] == C
        Hs, Ws = add_src.shape[1], add_src.shape[2]
        H, W = add_hw
    check(lib.uenc_groupnorm_tokens_fwd(x.data_ptr(), dt(x), gamma.data_ptr(), beta.data_ptr(), y.data_ptr(), dt(y), stats.data_ptr(),
                                        scratch.data_ptr(), ptr(add_src), Hs, Ws, H, W, B, HW, C, G, float(eps), int(relu),
                                        stream_ptr()), "groupnorm_tokens_fwd")
    return y, stats


def groupnorm_tokens_bwd(dy, x, gamma, beta, stats, G: int, *, relu=False, dgamma=None, dbeta=None, dx_dtype=torch.bfloat16):
    B, HW, C = x.shape
    assert dy.is_contiguous() and dy.shape == x.shape and x.is_contiguous()
    dx = torch.empty((B, HW, C), dtype=_odt(dx_dtype), device=x.device)
    scratch = _scratch("gn", int(lib.uenc_groupnorm_tokens_scratch_bytes(B, HW, C, G)), x.device)
    check(lib.uenc_groupnorm_tokens_bwd(dy.data_ptr(), dt(dy), x.data_ptr(), dt(x), gamma.data_ptr(), beta.data_ptr(), stats.data_ptr(),
                                        dx.data_ptr(), dt(dx), ptr(dgamma), ptr(dbeta), scratch.data_ptr(), B, HW, C, G, int(relu),
                                        stream_ptr()), "groupnorm_tokens_bwd")
    return dx


def upsample_bilinear_tokens_bwd(dy, Hs: int, Ws: int):
    """dy (B, H, W, C) fp32|bf16 -> (B, Hs, Ws, C) fp32: adjoint of the align_corners=False bilinear resize (Hs, Ws) -> (H, W)."""
    B, H, W, C = dy.shape
    assert dy.is_contiguous()
    out = torch.empty((B, Hs, Ws, C), dtype=torch.float32, device=dy.device)
    check(lib.uenc_upsample_bilinear_tokens_bwd(dy.data_ptr(), dt(dy), out.data_ptr(), B, H, W, Hs, Ws, C, stream_ptr()),
          "upsample_bilinear_tokens_bwd")
    return out


def im2col3x3(x16):
    """(B, H, W, C) bf16 -> (B*H*W, 9*C) bf16 patch matrix, column order (ky, kx, c)."""
    B, H, W, C = x16.shape
    if EXACT and x16.dtype == torch.float32:
        # pure data movement: gather the fp32 map as a bf16 map with twice the channels (same bytes, same (ky, kx, c) column order)
        return im2col3x3(x16.contiguous().view(torch.bfloat16)).view(torch.float32)
    assert x16.dtype == torch.bfloat16 and x16.is_contiguous()
    col = torch.empty((B * H * W, 9 * C), dtype=torch.bfloat16, device=x16.device)
    check(lib.uenc_im2col3x3(x16.data_ptr(), col.data_ptr(), B, H, W, C, stream_ptr()), "im2col3x3")
    return col


def col2im3x3(dcol, B: int, H: int, W: int, C: int):
    if EXACT and dcol.dtype == torch.float32:   # adjoint of the gather as nine shifted fp32 adds (verification mode only)
        d = dcol.view(B, H, W, 9, C)
        dxp = dcol.new_zeros((B, H + 2, W + 2, C))
        for t in range(9):
            ky, kx = divmod(t, 3)
            dxp[:, ky:ky + H, kx:kx + W] += d[:, :, :, t]
        return dxp[:, 1:1 + H, 1:1 + W].contiguous()
    assert dcol.dtype == torch.bfloat16 and dcol.is_contiguous() and dcol.numel() == B * H * W * 9 * C
    dx = torch.empty((B, H, W, C), dtype=torch.bfloat16, device=dcol.device)
    check(lib.uenc_col2im3x3(dcol.data_ptr(), dx.data_ptr(), B, H, W, C, stream_ptr()), "col2im3x3")
    return dx


# --------------------------------------------------------------------------------------------
# deformable encoder layer glue
# --------------------------------------------------------------------------------------------
def add_cast_bf16(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    """bf16(a + b); b may be one period of a (e.g. (S, C) against (B, S, C))."""
    assert a.dtype == torch.float32 and b.dtype == torch.float32 and a.is_contiguous() and b.is_contiguous()
    if EXACT:
        return (a.view(-1, b.numel()) + b.view(1, -1)).view(a.shape)
    out = torch.empty(a.shape, dtype=torch.bfloat16, device=a.device)
    check(lib.uenc_add_cast_bf16(a.data_ptr(), b.data_ptr(), out.data_ptr(), a.numel(), b.numel(), stream_ptr()), "add_cast_bf16")
    return out


def msda_prep_fwd(offaw: torch.Tensor, ref: torch.Tensor, shapes: torch.Tensor, B: int, Lq: int, M: int, L: int, P: int):
    """offaw (B*Lq, >= 3*M*L*P) fp32 -> loc (B, Lq, M, L, P, 2), aw (B, Lq, M, L, P) (softmaxed)."""
    assert offaw.dtype == torch.float32 and offaw.stride(1) == 1 and ref.dtype == torch.float32 and ref.is_contiguous()
    loc = torch.empty((B, Lq, M, L, P, 2), dtype=torch.float32, device=offaw.device)
    aw = torch.empty((B, Lq, M, L, P), dtype=torch.float32, device=offaw.device)
    check(lib.uenc_msda_prep_fwd(offaw.data_ptr(), offaw.stride(0), ref.data_ptr(), int(ref.shape[0] != 1), shapes.data_ptr(),
                                 loc.data_ptr(), aw.data_ptr(), B * Lq, Lq, M, L, P, stream_ptr()), "msda_prep_fwd")
    return loc, aw


def msda_prep_bwd(dloc, daw, aw, shapes, ncols: int):
    """-> d(offaw) (B*Lq, ncols) bf16."""
    B, Lq, M, L, P, _ = dloc.shape
    assert dloc.is_contiguous() and daw.is_contiguous() and aw.is_contiguous() and ncols == 3 * M * L * P
    if EXACT:   # fp32 result: d(offset) = d(loc) / (W_l, H_l); d(logit) = aw * (d(aw) - sum_{l,p} aw * d(aw))   (elementwise glue)
        norm = torch.stack([shapes[:, 1], shapes[:, 0]], -1).to(torch.float32).view(1, 1, 1, L, 1, 2)
        doff = (dloc / norm).reshape(B * Lq, 2 * M * L * P)
        a, g = aw.view(B * Lq, M, L * P), daw.view(B * Lq, M, L * P)
        dlog = (a * (g - (a * g).sum(-1, keepdim=True))).reshape(B * Lq, M * L * P)
        return torch.cat([doff, dlog], 1).contiguous()
    out = torch.empty((B * Lq, ncols), dtype=torch.bfloat16, device=dloc.device)
    check(lib.uenc_msda_prep_bwd(dloc.data_ptr(), daw.data_ptr(), aw.data_ptr(), shapes.data_ptr(), out.data_ptr(), ncols, B * Lq, Lq,
                                 M, L, P, stream_ptr()), "msda_prep_bwd")
    return out


def segment_colsum(x16: torch.Tensor, seg_start: torch.Tensor, rows_per_image: int, images: int) -> torch.Tensor:
    """x16 (images * rows_per_image, cols) bf16 -> (nseg, cols) fp32 sums over the row segments of every image."""
    if EXACT:
        st = seg_start.tolist() + [rows_per_image]
        xv = x16.view(images, rows_per_image, -1)
        return torch.stack([xv[:, st[i]:st[i + 1]].sum((0, 1)) for i in range(len(st) - 1)])
    assert x16.dtype == torch.bfloat16 and x16.stride(1) == 1 and seg_start.dtype == torch.int64
    nseg, cols = seg_start.numel(), x16.shape[1]
    out = torch.empty((nseg, 128, cols), dtype=torch.float32, device=x16.device)          # 128 stored block partials per segment
    check(lib.uenc_segment_colsum(x16.data_ptr(), x16.stride(0), cols, seg_start.data_ptr(), nseg, rows_per_image, images, out.data_ptr(),
                                  stream_ptr()), "segment_colsum")
    return out.sum(1)
