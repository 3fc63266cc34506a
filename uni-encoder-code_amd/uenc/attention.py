"""Decoder multi-head attention core on the HIP kernels of csrc/mha.hip.

q (B, Lq, E), k / v (B, S, E) bf16 with heads interleaved in E (head_dim 32), any row / batch strides
(slices of packed in-projection outputs are read in place); mask (B, Lq, S) bool, True = blocked, shared
by all heads.  Forward = split-KV flash attention + combine; backward = dQ kernel + dK/dV kernel.
"""
from typing import Optional

import torch

from . import kernels as K
from .capi import check, lib, stream_ptr


def _strided(t: torch.Tensor) -> torch.Tensor:
    """(B, L, E) with unit inner stride and 16-byte aligned rows, copying only if necessary."""
    if t.dtype != K.adt():
        t = t.to(K.adt())
    if t.stride(2) != 1 or t.stride(1) % 8 or t.stride(0) % 8 or t.data_ptr() % 16:
        t = t.contiguous()
    return t


def _mask_bytes(mask: Optional[torch.Tensor]):
    if mask is None:
        return None, 0
    m = mask.to(torch.uint8) if mask.dtype != torch.bool else mask.view(torch.uint8)
    S = m.shape[-1]
    if S % 4 or not m.is_contiguous():
        Sp = -(-S // 4) * 4
        m = torch.nn.functional.pad(m, (0, Sp - S)).contiguous()
    return m, m.shape[-1]


class MHACoreFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, q, k, v, nheads, mask, dropout_p=0.0, seed=0):
        q, k, v = _strided(q), _strided(k), _strided(v)
        B, Lq, E = q.shape
        S = k.shape[1]
        assert E == nheads * 32, "the HIP attention kernels are built for head_dim 32"
        scale = 32 ** -0.5
        m8, mrs = _mask_bytes(mask)
        lse = torch.empty((B, nheads, Lq), dtype=torch.float32, device=q.device)
        if K.EXACT:                                      # fp32 operands, csrc/exact.hip
            out = torch.empty((B, Lq, E), dtype=torch.float32, device=q.device)
            check(lib.uenc_mha_f32_fwd(q.data_ptr(), q.stride(0), q.stride(1), k.data_ptr(), k.stride(0), k.stride(1),
                                       v.data_ptr(), v.stride(0), v.stride(1), m8.data_ptr() if m8 is not None else 0, mrs,
                                       out.data_ptr(), out.stride(0), out.stride(1), lse.data_ptr(), B, nheads, Lq, S, scale, float(dropout_p), int(seed),
                                       stream_ptr()), "mha_f32_fwd")
            ctx.save_for_backward(q, k, v, out, lse, m8 if m8 is not None else torch.empty(0, device=q.device))
            ctx.meta = (nheads, S, scale, mrs, m8 is not None, float(dropout_p), int(seed))
            return out
        out = torch.empty((B, Lq, E), dtype=torch.bfloat16, device=q.device)
        nws = lib.uenc_mha_fwd_workspace_floats(B, nheads, Lq, S)
        ws = torch.empty((nws,), dtype=torch.float32, device=q.device) if nws else None
        check(lib.uenc_mha_fwd(q.data_ptr(), q.stride(0), q.stride(1), k.data_ptr(), k.stride(0), k.stride(1),
                               v.data_ptr(), v.stride(0), v.stride(1), m8.data_ptr() if m8 is not None else 0, mrs,
                               out.data_ptr(), out.stride(0), out.stride(1), lse.data_ptr(),
                               ws.data_ptr() if ws is not None else 0, B, nheads, Lq, S, scale, float(dropout_p), int(seed), stream_ptr()),
              "mha_fwd")
        ctx.save_for_backward(q, k, v, out, lse, m8 if m8 is not None else torch.empty(0, device=q.device))
        ctx.meta = (nheads, S, scale, mrs, m8 is not None, float(dropout_p), int(seed))
        return out

    @staticmethod
    def backward(ctx, dout):
        q, k, v, out, lse, m8 = ctx.saved_tensors
        nheads, S, scale, mrs, has_mask, dropout_p, seed = ctx.meta
        B, Lq, E = q.shape
        dout = _strided(dout)
        if K.EXACT:
            dq = torch.empty((B, Lq, E), dtype=torch.float32, device=q.device)
            dk = torch.zeros((B, S, E), dtype=torch.float32, device=q.device)
            dv = torch.zeros((B, S, E), dtype=torch.float32, device=q.device)
            delta = torch.empty((B, nheads, Lq), dtype=torch.float32, device=q.device)
            check(lib.uenc_mha_f32_bwd(q.data_ptr(), q.stride(0), q.stride(1), k.data_ptr(), k.stride(0), k.stride(1),
                                       v.data_ptr(), v.stride(0), v.stride(1), m8.data_ptr() if has_mask else 0, mrs,
                                       out.data_ptr(), out.stride(0), out.stride(1), lse.data_ptr(),
                                       dout.data_ptr(), dout.stride(0), dout.stride(1),
                                       dq.data_ptr(), dq.stride(0), dq.stride(1), dk.data_ptr(), dk.stride(0), dk.stride(1),
                                       dv.data_ptr(), dv.stride(0), dv.stride(1), delta.data_ptr(), B, nheads, Lq, S, scale, dropout_p, seed,
                                       stream_ptr()), "mha_f32_bwd")
            return dq, dk, dv, None, None, None, None
        dq = torch.zeros((B, Lq, E), dtype=torch.float32, device=q.device)
        dk = torch.empty((B, S, E), dtype=torch.bfloat16, device=q.device)
        dv = torch.empty((B, S, E), dtype=torch.bfloat16, device=q.device)
        check(lib.uenc_mha_bwd(q.data_ptr(), q.stride(0), q.stride(1), k.data_ptr(), k.stride(0), k.stride(1),
                               v.data_ptr(), v.stride(0), v.stride(1), m8.data_ptr() if has_mask else 0, mrs,
                               out.data_ptr(), out.stride(0), out.stride(1), lse.data_ptr(),
                               dout.data_ptr(), dout.stride(0), dout.stride(1),
                               dq.data_ptr(), dq.stride(0), dq.stride(1), dk.data_ptr(), dk.stride(0), dk.stride(1),
                               dv.data_ptr(), dv.stride(0), dv.stride(1), B, nheads, Lq, S, scale, dropout_p, seed, stream_ptr()), "mha_bwd")
        return dq.to(torch.bfloat16), dk, dv, None, None, None, None


def mha(q, k, v, nheads: int, mask: Optional[torch.Tensor] = None, dropout_p: float = 0.0, seed: int = 0):
    """dropout_p / seed: dropout on the attention probabilities (training mode of nn.MultiheadAttention(dropout=p)): the kernels
    derive the keep-mask from a hash of (seed, element index), so the backward regenerates it and nothing is stored."""
    if not q.is_cuda:
        raise RuntimeError("uenc attention runs on the GPU only")
    return MHACoreFn.apply(q, k, v, nheads, mask, dropout_p, seed)


def keep_mask_reference(B: int, nheads: int, Lq: int, S: int, dropout_p: float, seed: int) -> torch.Tensor:
    """(B, nheads, Lq, S) bool: the keep-mask the kernels derive (common.h attn_keep), restated with int64 torch arithmetic on the
    host.  Test helper: lets a dense PyTorch attention reproduce a dropped-out kernel result exactly."""
    idx = torch.arange(B * nheads * Lq * S, dtype=torch.int64)
    M = 0xFFFFFFFF
    x = (((idx & M) ^ (((idx >> 32) * 0x9E3779B9) & M)) + seed) & M
    x = x ^ (x >> 16); x = (x * 0x7feb352d) & M
    x = x ^ (x >> 15); x = (x * 0x846ca68b) & M
    x = x ^ (x >> 16)
    thresh = 0 if dropout_p <= 0 else int(float(torch.tensor(dropout_p, dtype=torch.float32)) * 4294967296.0)
    return (x >= thresh).view(B, nheads, Lq, S)
