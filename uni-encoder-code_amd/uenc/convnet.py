"""Convolution layers of the "sequence" (depth / pose / motion) branch on the HIP GEMMs (inference).

The branch's decoders (reference model/modeling/pose_decoder/resnet_like_pose_decoder.py, motion_decoder/dynamo_motion_decoder_mod.py,
pixel_decoder/transdssl.py) are plain conv nets: 1x1 and 3x3 (stride 1 / 2, padding 1) convolutions, eval-mode BatchNorm, ReLU / ELU,
bilinear resizes.  Here every convolution is the library's bf16 MFMA GEMM: 1x1 = a GEMM over the channels-last map, 3x3 = HIP patch
gather (`uenc_im2col3x3`, `uenc_im2col3x3_s2`) + GEMM, with the bias, the eval-mode BatchNorm (folded into the weight rows and the
bias: y = s * conv(x) + t) and a following ReLU in the GEMM epilogue.  Maps stay channels-last ((B, C, H, W)-shaped views of
(B, H, W, C) storage) so the ATen resizes / concatenations between the layers copy nothing extra.

Forward only (the reference drives this branch in eval mode only: train_net.py:283 asserts --eval-only and the branch has no loss).
"""
from typing import Optional

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import kernels as K
from . import ops


def _bn_affine(bn: Optional[nn.BatchNorm2d]):
    if bn is None:
        return None, None
    if bn.training:
        raise NotImplementedError("the sequence-branch decoders run in eval mode only (BatchNorm with running statistics)")
    s = bn.weight.detach() * torch.rsqrt(bn.running_var + bn.eps)
    return s, bn.bias.detach() - bn.running_mean * s


def _operands(conv: nn.Conv2d, bn: Optional[nn.BatchNorm2d]):
    """(W2 (Np, Kp) operand in (ky, kx, c) column order with channels padded to 8 and the BatchNorm scale folded in, bias (Np) fp32);
    cached while the parameters / statistics are unchanged."""
    w = conv.weight
    tensors = [w] + ([conv.bias] if conv.bias is not None else []) + ([bn.weight, bn.bias, bn.running_mean, bn.running_var] if bn is not None else [])
    key = tuple((t.data_ptr(), t._version) for t in tensors) + (K.EXACT,)
    ent = getattr(conv, "_uenc_ops", None)
    if ent is not None and ent[0] == key:
        return ent[1], ent[2]
    Co, Ci, kh, kw = w.shape
    Cp, Np = -(-Ci // 8) * 8, -(-Co // 8) * 8
    s, t = _bn_affine(bn)
    wf = w.detach().float()
    if s is not None:
        wf = wf * s.view(-1, 1, 1, 1)
    w2 = torch.zeros((Np, kh, kw, Cp), dtype=torch.float32, device=w.device)
    w2[:Co, :, :, :Ci] = wf.permute(0, 2, 3, 1)
    b = torch.zeros((Np,), dtype=torch.float32, device=w.device)
    if conv.bias is not None:
        b[:Co] = conv.bias.detach().float() * (s if s is not None else 1.0)
    if t is not None:
        b[:Co] += t
    w2 = K.cast_bf16(w2.view(Np, kh * kw * Cp).contiguous())
    conv._uenc_ops = (key, w2, b)
    return w2, b


def conv_bn_act(x: torch.Tensor, conv: nn.Conv2d, bn: Optional[nn.BatchNorm2d] = None, relu: bool = False, out_dtype=torch.float32) -> torch.Tensor:
    """act(bn(conv(x))) for x (B, C, H, W) (any strides; channels-last is free) -> (B, Cout, H', W') channels-last view.
    Supported: 1x1 (stride 1 / 2, no padding) and 3x3 (stride 1 / 2, padding 1), groups 1, dilation 1."""
    assert conv.groups == 1 and conv.dilation == (1, 1)
    B, C, H, W = x.shape
    Co = conv.out_channels
    kh, kw = conv.kernel_size
    st = conv.stride[0]
    assert conv.stride in ((1, 1), (2, 2))
    w2, b = _operands(conv, bn)
    Cp = w2.shape[1] // (kh * kw)
    xt = x.permute(0, 2, 3, 1)                                          # (B, H, W, C)
    if (kh, kw) == (1, 1):
        assert conv.padding == (0, 0)
        if st == 2:
            xt = xt[:, ::2, ::2]
        Ho, Wo = xt.shape[1], xt.shape[2]
        a = xt.to(K.adt())
        if Cp != C:
            a = F.pad(a, (0, Cp - C))
        a = a.reshape(B * Ho * Wo, Cp)
        a = a if a.is_contiguous() else a.contiguous()
    else:
        assert (kh, kw) == (3, 3) and conv.padding == (1, 1)
        a = xt.to(K.adt())
        if Cp != C:
            a = F.pad(a, (0, Cp - C))
        a = a if a.is_contiguous() else a.contiguous()
        if st == 1:
            Ho, Wo = H, W
            a = K.im2col3x3(a)
        else:
            Ho, Wo = (H + 1) // 2, (W + 1) // 2
            a = K.im2col3x3_s2(a)
    out = K.gemm_nt(a, w2, bias=b, epilogue=K.EPI_RELU if relu else K.EPI_NONE, out_dtype=out_dtype)
    if out.shape[1] != Co:
        out = out[:, :Co]
    return out.reshape(B, Ho, Wo, Co).permute(0, 3, 1, 2)


def conv(x, c: nn.Conv2d, relu: bool = False, out_dtype=torch.float32):
    return conv_bn_act(x, c, None, relu, out_dtype)
