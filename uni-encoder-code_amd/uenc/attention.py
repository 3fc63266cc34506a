"""Decoder multi-head attention core (self-attention over 150 queries, masked cross-attention over
2k-32k keys, class-transformer cross-attention over 131k keys).

TRANSITIONAL: until `csrc/mha.hip` lands this routes through torch's ROCm SDPA on the GPU (never a
CPU path).  DESIGN.md lists it under "not yet HIP".
"""
from typing import Optional

import torch
import torch.nn.functional as F


def mha(q, k, v, nheads: int, mask: Optional[torch.Tensor] = None):
    B, Lq, E = q.shape
    S = k.shape[1]
    hd = E // nheads
    if not q.is_cuda:
        raise RuntimeError("uenc attention runs on the GPU only")
    qh = q.reshape(B, Lq, nheads, hd).transpose(1, 2)
    kh = k.reshape(B, S, nheads, hd).transpose(1, 2)
    vh = v.reshape(B, S, nheads, hd).transpose(1, 2)
    am = None if mask is None else ~mask[:, None]
    o = F.scaled_dot_product_attention(qh, kh, vh, attn_mask=am)
    return o.transpose(1, 2).reshape(B, Lq, E)
