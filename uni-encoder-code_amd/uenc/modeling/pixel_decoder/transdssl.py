"""`TransDSSL` depth decoder of the "sequence" branch -- counterpart of reference model/modeling/pixel_decoder/transdssl.py:10-404,
registered as `SEM_SEG_HEADS_REGISTRY["TransDSSL"]` (cfg `MODEL.SEM_SEG_HEAD.DEPTH_DECODER_NAME`, oneformer_R50_bs16_90k.yaml:14).
Same module tree / state-dict names (`layers.layer{1-4}_rn`, `layers.refinenet{0-4}.{resConfUnit1,resConfUnit2,en_atten,out_conv}`,
`layers.output_conv{,2,3,4}`), same `forward_features(features) -> {("disp", s): (B, 1, H/2^s, W/2^s)}`; input channels hard-wired to
Swin-T (:332-334).  Convolutions on the HIP GEMMs (uenc/convnet.py); resizes / softmax / sums are ATen on channels-last maps."""
from typing import Dict

import torch
import torch.nn as nn
import torch.nn.functional as F

from ...convnet import conv, conv_bn_act
from ...d2 import SEM_SEG_HEADS_REGISTRY, ShapeSpec


def _make_1x1_convs(in_shape, out_shape, groups=1, expand=False):
    out = nn.Module()
    shapes = [out_shape, out_shape * 2, out_shape * 4, out_shape * 8] if expand else [out_shape] * 4
    for i in range(4):
        setattr(out, f"layer{i + 1}_rn", nn.Conv2d(in_shape[i], shapes[i], kernel_size=1, stride=1, padding=0, bias=False, groups=groups))
    return out


class Interpolate(nn.Module):
    def __init__(self, scale_factor, mode, align_corners=False):
        super().__init__()
        self.scale_factor, self.mode, self.align_corners = scale_factor, mode, align_corners

    def forward(self, x):
        return F.interpolate(x, scale_factor=self.scale_factor, mode=self.mode, align_corners=self.align_corners)


class ResidualConvUnit(nn.Module):
    def __init__(self, features, activation, layer_norm, isFlow=False):
        super().__init__()
        self.layer_norm, self.isFlow, self.groups = layer_norm, isFlow, 1
        self.conv1 = nn.Conv2d(features, features, kernel_size=3, stride=1, padding=1, bias=not layer_norm, groups=1)
        self.conv2 = nn.Conv2d(features, features, kernel_size=3, stride=1, padding=1, bias=not layer_norm, groups=1)
        if layer_norm:
            self.layer_norm1 = nn.BatchNorm2d(features)
            self.layer_norm2 = nn.BatchNorm2d(features)
        self.activation = activation

    def forward(self, x):
        out = conv_bn_act(self.activation(x), self.conv1, self.layer_norm1 if self.layer_norm else None, relu=isinstance(self.activation, nn.ReLU),
                          out_dtype=torch.bfloat16)
        if not isinstance(self.activation, nn.ReLU):
            out = self.activation(out)
        out = conv_bn_act(out, self.conv2, self.layer_norm2 if self.layer_norm else None)
        return out + x


class SoftAttDepth(nn.Module):
    """Expectation of a uniform ('UD') or log-spaced ('SID') depth grid under the channel softmax (:187-222)."""

    def __init__(self, alpha=0.01, beta=1.0, dim=1, discretization="UD"):
        super().__init__()
        self.dim, self.alpha, self.beta, self.discretization = dim, alpha, beta, discretization

    def forward(self, input_t, eps=1e-6):
        depth = input_t.shape[1]
        if self.discretization == "SID":
            k = torch.arange(depth, dtype=torch.float32)
            grid = torch.exp(torch.log(torch.tensor(self.alpha)) + torch.log(torch.tensor(self.beta / self.alpha)) * k / depth)
        else:
            grid = torch.linspace(self.alpha, self.beta, depth)
        z = F.softmax(input_t.float(), dim=self.dim) * grid.to(input_t.device).view(1, -1, 1, 1)
        return torch.sum(z, dim=1, keepdim=True)


class FeatureFusionBlock_custom(nn.Module):
    def __init__(self, features, activation, deconv=False, layer_norm=False, expand=False, align_corners=True, scale=1, input_length=2):
        super().__init__()
        self.deconv, self.align_corners, self.scale, self.groups, self.expand = deconv, align_corners, scale, 1, expand
        out_features = features if (not expand or features == 256) else features // 2
        self.out_conv = nn.Conv2d(features, out_features, kernel_size=1, stride=1, padding=0, bias=True, groups=1)
        self.dim = 1
        if input_length == 2:
            self.resConfUnit1 = ResidualConvUnit(features, activation, layer_norm)
            self.en_atten = nn.Conv2d(in_channels=features, out_channels=features, kernel_size=1, stride=1, padding=0)
        self.resConfUnit2 = ResidualConvUnit(features, activation, layer_norm)

    def forward(self, *xs):
        df = xs[0]
        if len(xs) == 2:
            if self.scale != 1:
                raise NotImplementedError("scale != 1 drops into a debugger in the reference (transdssl.py:283-285)")
            res = df + xs[1]
            att = F.softmax(conv(self.resConfUnit1(xs[1]), self.en_atten), dim=self.dim)
            output = self.resConfUnit2(res * att) + res
        else:
            output = self.resConfUnit2(df)
        output = F.interpolate(output, scale_factor=2, mode="bilinear", align_corners=self.align_corners)
        return conv(output, self.out_conv)


def _make_fusion_block(features, use_norm, scale=1, input_length=2):
    return FeatureFusionBlock_custom(features, nn.ReLU(False), deconv=False, layer_norm=use_norm, expand=False, align_corners=True, scale=scale,
                                     input_length=input_length)


@SEM_SEG_HEADS_REGISTRY.register()
class TransDSSL(nn.Module):
    def __init__(self, cfg, input_shape: Dict[str, ShapeSpec] = None, *, features=256, use_norm=False):
        super().__init__()
        self.layers = _make_1x1_convs([96, 192, 384, 768, 48], features, groups=1, expand=False)
        self.upsample = Interpolate(scale_factor=2, mode="bilinear", align_corners=True)
        for i, n in ((0, 2), (1, 2), (2, 2), (3, 2), (4, 1)):
            setattr(self.layers, f"refinenet{i}", _make_fusion_block(features, use_norm, input_length=n))
        self.attn_depth = SoftAttDepth()
        for name in ("output_conv4", "output_conv3", "output_conv2", "output_conv"):
            setattr(self.layers, name, nn.Sequential(nn.Conv2d(features, features // 2, kernel_size=3, stride=1, padding=1),
                                                     nn.Conv2d(features // 2, 32, kernel_size=3, stride=1, padding=1)))

    @classmethod
    def from_config(cls, cfg, input_shape):
        return {"input_shape": {k: v for k, v in input_shape.items() if k in cfg.MODEL.SEM_SEG_HEAD.IN_FEATURES}}

    def _disp(self, head, x):
        return self.attn_depth(conv(conv(x, head[0], out_dtype=torch.bfloat16), head[1]))

    def forward_features(self, features):
        L = self.layers
        layer_1_rn = conv(features["res2"], L.layer1_rn)
        layer_2_rn = conv(features["res3"], L.layer2_rn)
        layer_3_rn = conv(features["res4"], L.layer3_rn)
        layer_4_rn = conv(features["res5"], L.layer4_rn)
        path_4 = L.refinenet4(layer_4_rn)
        path_3 = L.refinenet3(path_4, layer_3_rn)
        disp_3 = self._disp(L.output_conv4, path_3)
        path_2 = L.refinenet2(path_3, layer_2_rn)
        disp_2 = self._disp(L.output_conv3, path_2)
        path_1 = L.refinenet1(path_2, layer_1_rn)
        disp_1 = self._disp(L.output_conv2, path_1)
        layer_0_rn = self.upsample(layer_1_rn)
        path_0 = L.refinenet0(path_1, layer_0_rn)
        disp_0 = self._disp(L.output_conv, path_0)
        return {("disp", 3): disp_3, ("disp", 2): disp_2, ("disp", 1): disp_1, ("disp", 0): disp_0}
