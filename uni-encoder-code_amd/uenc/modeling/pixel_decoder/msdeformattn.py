"""MSDeformAttn pixel decoder — counterpart of reference model/modeling/pixel_decoder/msdeformattn.py.

Same class names, constructor arguments and parameter names (`input_proj.{i}.{0,1}`,
`transformer.level_embed`, `transformer.encoder.layers.{l}.{self_attn,linear1,linear2,norm1,norm2}`,
`mask_features`, `adapter_1`, `layer_1`), same `forward_features(features) -> (mask_features,
encoder_out[0], multi_scale_features)` contract, registered as `MSDeformAttnPixelDecoder`.

Tokens stay channels-last `(B, sum(HW), 256)` fp32 through the encoder; every Linear / 1x1 conv is a
bf16 MFMA GEMM with fused bias / ReLU / residual epilogues; LayerNorm, GroupNorm (channels-last, with the
FPN top-down merge and ReLU folded in), the 3x3 FPN conv (patch gather + GEMM), the 12-way softmax with the
sampling-location arithmetic (`msda_prep`) and the deformable sampling core are HIP kernels.  ATen remains
only for a GroupNorm shape outside the kernels' domain and a non-GroupNorm `norm` (no reference config has
either; tests/test_model_gpu.py::test_pixel_decoder_unfused_side_paths drives those arms).
"""
from typing import Callable, Dict, List, Optional, Union

import numpy as np
import os

import torch
import torch.nn as nn
import torch.nn.functional as F

from ... import ops
from ...d2 import SEM_SEG_HEADS_REGISTRY, Conv2d, ShapeSpec, configurable, get_norm
from ..transformer_decoder.position_encoding import PositionEmbeddingSine
from .ops import MSDeformAttn


def _tokens(x: torch.Tensor) -> torch.Tensor:
    """(B, C, H, W) -> (B, H*W, C); free when x is stored channels-last (as the backbone emits)."""
    B, C, H, W = x.shape
    return x.permute(0, 2, 3, 1).reshape(B, H * W, C)


def _conv1x1_gn(tok: torch.Tensor, conv: nn.Module, gn: Optional[nn.Module], H: int, W: int, **gn_kw) -> torch.Tensor:
    """1x1 conv (GEMM) + GroupNorm on channels-last tokens -> (B, HW, D); gn_kw: see ops.group_norm_tokens."""
    y = ops.linear(tok, conv.weight, conv.bias, out_dtype=torch.float32)
    if gn is None:
        return y
    if isinstance(gn, nn.GroupNorm) and _gn_tokens_ok(gn):
        return ops.group_norm_tokens(y, gn, **gn_kw)
    assert not gn_kw
    B, HW, D = y.shape
    y = F.group_norm(y.transpose(1, 2).reshape(B, D, H, W), gn.num_groups, gn.weight, gn.bias, gn.eps)
    return y.flatten(2).transpose(1, 2)


def _gn_tokens_ok(gn: nn.Module) -> bool:
    """Shapes the channels-last GroupNorm kernels take (the reference's GN(32, 256) does)."""
    C, G = gn.num_channels, gn.num_groups
    return gn.affine and C % G == 0 and (C // G) % 4 == 0 and 64 % (C // G) == 0 and C <= 1024 and 256 % (C // 4) == 0


class MSDeformAttnTransformerEncoderLayer(nn.Module):
    def __init__(self, d_model=256, d_ffn=1024, dropout=0.1, activation="relu", n_levels=4, n_heads=8, n_points=4):
        super().__init__()
        assert activation == "relu"
        self.self_attn = MSDeformAttn(d_model, n_levels, n_heads, n_points)
        self.norm1 = nn.LayerNorm(d_model)
        self.linear1 = nn.Linear(d_model, d_ffn)
        self.linear2 = nn.Linear(d_ffn, d_model)
        self.norm2 = nn.LayerNorm(d_model)
        self.dropout_p = dropout            # dropout1 / dropout2 / dropout3 of the reference (msdeformattn.py:111-119), training mode only
        self._masks = []                    # the keep-masks of the last UNFUSED training forward, in order (for tests)
        self._seeds = None                  # the three dropout seeds of the last fused training forward (for tests)

    def _drop(self, t):
        """Inverted dropout with an explicit keep-mask (torch.nn.Dropout semantics: x * mask / keep_prob)."""
        keep = 1.0 - self.dropout_p
        mask = torch.rand(t.shape, device=t.device) < keep
        self._masks.append(mask)
        return t * (mask.to(t.dtype) / keep)

    def forward(self, src, pos, reference_points, spatial_shapes, level_start_index, padding_mask=None, level_embed=None):
        a = self.self_attn
        fusable = (level_embed is not None and pos is not None and padding_mask is None and reference_points.shape[-1] == 2
                   and a.d_model % 8 == 0 and a.d_model // a.n_heads in (16, 32, 64) and a.n_levels * a.n_points <= 16)
        drop = None
        if self.training and self.dropout_p > 0.0 and fusable and not ops.is_exact():
            # three seeds from torch's CPU generator (no device sync); the kernels derive the keep-masks from (seed, element index)
            self._seeds = tuple(int(v) for v in torch.randint(0, 2 ** 31 - 1, (3,)).tolist())
            drop = (float(self.dropout_p),) + self._seeds
        elif self.training and self.dropout_p > 0.0:
            # training with dropout: the layer op by op (same kernels), masks applied between them as in the reference (:121-142)
            self._masks = []
            q = src if pos is None else src + pos
            h = self.self_attn(q, reference_points, src, spatial_shapes, level_start_index, padding_mask)
            src = ops.layer_norm(src + self._drop(h.float()), self.norm1.weight, self.norm1.bias)
            t = self._drop(F.relu(ops.linear(src, self.linear1.weight, self.linear1.bias)))
            t = ops.linear(t, self.linear2.weight, self.linear2.bias)
            return ops.layer_norm(src + self._drop(t.float()), self.norm2.weight, self.norm2.bias)
        if fusable:
            # the whole layer as one autograd node (ops.DeformEncoderLayerFn); `pos` is then a constant and the gradient of
            # the level embedding inside it is returned through `level_embed`
            return ops.deform_encoder_layer(
                src, pos, level_embed, reference_points, spatial_shapes, level_start_index, a.n_heads, a.n_points,
                [a.value_proj.weight, a.value_proj.bias, a.sampling_offsets.weight, a.sampling_offsets.bias,
                 a.attention_weights.weight, a.attention_weights.bias, a.output_proj.weight, a.output_proj.bias,
                 self.norm1.weight, self.norm1.bias, self.linear1.weight, self.linear1.bias, self.linear2.weight, self.linear2.bias,
                 self.norm2.weight, self.norm2.bias], drop=drop)
        q = src if pos is None else src + pos
        h = self.self_attn(q, reference_points, src, spatial_shapes, level_start_index, padding_mask, residual=src)
        src = ops.layer_norm(h, self.norm1.weight, self.norm1.bias)
        h = ops.mlp(src, [self.linear1.weight, self.linear1.bias, self.linear2.weight, self.linear2.bias],
                    act="relu", residual=src)
        return ops.layer_norm(h, self.norm2.weight, self.norm2.bias)


class MSDeformAttnTransformerEncoder(nn.Module):
    def __init__(self, encoder_layer_args, num_layers):
        super().__init__()
        self.layers = nn.ModuleList([MSDeformAttnTransformerEncoderLayer(*encoder_layer_args) for _ in range(num_layers)])
        self.num_layers = num_layers

    @staticmethod
    def get_reference_points(spatial_shapes, valid_ratios, device):
        pts = []
        for lvl, (H_, W_) in enumerate(spatial_shapes):
            ref_y, ref_x = torch.meshgrid(torch.linspace(0.5, H_ - 0.5, H_, dtype=torch.float32, device=device),
                                          torch.linspace(0.5, W_ - 0.5, W_, dtype=torch.float32, device=device), indexing="ij")
            ref_y = ref_y.reshape(-1)[None] / (valid_ratios[:, None, lvl, 1] * H_)
            ref_x = ref_x.reshape(-1)[None] / (valid_ratios[:, None, lvl, 0] * W_)
            pts.append(torch.stack((ref_x, ref_y), -1))
        ref = torch.cat(pts, 1)
        return ref[:, :, None] * valid_ratios[:, None]

    def forward(self, src, spatial_shapes, level_start_index, valid_ratios, pos=None, padding_mask=None, shapes_list=None, ref=None,
                level_embed=None):
        out = src
        if ref is None:
            ref = self.get_reference_points(shapes_list, valid_ratios, src.device).contiguous()
        for layer in self.layers:
            out = layer(out, pos, ref, spatial_shapes, level_start_index, padding_mask, level_embed)
        return out


class MSDeformAttnTransformerEncoderOnly(nn.Module):
    def __init__(self, d_model=256, nhead=8, num_encoder_layers=6, dim_feedforward=1024, dropout=0.1, activation="relu",
                 num_feature_levels=4, enc_n_points=4):
        super().__init__()
        self.d_model, self.nhead = d_model, nhead
        self.encoder = MSDeformAttnTransformerEncoder(
            (d_model, dim_feedforward, dropout, activation, num_feature_levels, nhead, enc_n_points), num_encoder_layers)
        self.level_embed = nn.Parameter(torch.Tensor(num_feature_levels, d_model))
        self._geometry = {}
        for p in self.parameters():
            if p.dim() > 1:
                nn.init.xavier_uniform_(p)
        for m in self.modules():
            if isinstance(m, MSDeformAttn):
                m._reset_parameters()
        nn.init.normal_(self.level_embed)

    def forward(self, srcs_tok, pos_tok, shapes_list):
        """srcs_tok / pos_tok: per-level (B, HW, D) channels-last tokens; shapes_list [(H, W)]."""
        device = srcs_tok[0].device
        B = srcs_tok[0].shape[0]
        src = torch.cat(srcs_tok, 1)
        a = self.encoder.layers[0].self_attn
        fused = a.d_model % 8 == 0 and a.d_model // a.n_heads in (16, 32, 64) and a.n_levels * a.n_points <= 16
        if self.training and self.encoder.layers[0].dropout_p > 0.0 and ops.is_exact():
            fused = False                       # fp32 verification mode: dropout through the composed path (explicit ATen masks)
        if fused:
            # pos = sine embedding + level embedding: a constant map for the fused layers (one period (S, C), shared by the
            # batch), which return the level embedding's gradient themselves
            with torch.no_grad():
                pos = torch.cat([p[:1] + self.level_embed[l].view(1, 1, -1) for l, p in enumerate(pos_tok)], 1)
        else:
            pos = torch.cat([p + self.level_embed[l].view(1, 1, -1) for l, p in enumerate(pos_tok)], 1)
        key = (tuple(shapes_list), B, str(device))
        geo = self._geometry.get(key)
        if geo is None:       # level geometry lives on the device; built once per shape (no per-step host-to-device copies)
            spatial_shapes = torch.as_tensor(shapes_list, dtype=torch.long, device=device)
            level_start_index = torch.cat((spatial_shapes.new_zeros((1,)), spatial_shapes.prod(1).cumsum(0)[:-1]))
            valid_ratios = torch.ones((B, len(shapes_list), 2), dtype=torch.float32, device=device)   # no padding masks here
            ref = MSDeformAttnTransformerEncoder.get_reference_points(shapes_list, valid_ratios, device).contiguous()
            geo = (spatial_shapes, level_start_index, valid_ratios, ref)
            self._geometry[key] = geo
        spatial_shapes, level_start_index, valid_ratios, ref = geo
        memory = self.encoder(src, spatial_shapes, level_start_index, valid_ratios, pos, None, shapes_list, ref,
                              self.level_embed if fused else None)
        return memory, spatial_shapes, level_start_index, valid_ratios


@SEM_SEG_HEADS_REGISTRY.register()
class MSDeformAttnPixelDecoder(nn.Module):
    @configurable
    def __init__(self, input_shape: Dict[str, ShapeSpec], *, transformer_dropout: float, transformer_nheads: int,
                 transformer_dim_feedforward: int, transformer_enc_layers: int, conv_dim: int, mask_dim: int,
                 norm: Optional[Union[str, Callable]] = None, transformer_in_features: List[str], common_stride: int):
        super().__init__()
        transformer_input_shape = {k: v for k, v in input_shape.items() if k in transformer_in_features}
        input_shape = sorted(input_shape.items(), key=lambda x: x[1].stride)
        self.in_features = [k for k, v in input_shape]
        self.feature_strides = [v.stride for k, v in input_shape]
        self.feature_channels = [v.channels for k, v in input_shape]
        transformer_input_shape = sorted(transformer_input_shape.items(), key=lambda x: x[1].stride)
        self.transformer_in_features = [k for k, v in transformer_input_shape]
        transformer_in_channels = [v.channels for k, v in transformer_input_shape]
        self.transformer_feature_strides = [v.stride for k, v in transformer_input_shape]
        self.transformer_num_feature_levels = len(self.transformer_in_features)
        chans = transformer_in_channels[::-1] if self.transformer_num_feature_levels > 1 else [transformer_in_channels[-1]]
        self.input_proj = nn.ModuleList([nn.Sequential(nn.Conv2d(c, conv_dim, kernel_size=1), nn.GroupNorm(32, conv_dim))
                                         for c in chans])
        for proj in self.input_proj:
            nn.init.xavier_uniform_(proj[0].weight, gain=1)
            nn.init.constant_(proj[0].bias, 0)
        self.transformer = MSDeformAttnTransformerEncoderOnly(
            d_model=conv_dim, dropout=transformer_dropout, nhead=transformer_nheads,
            dim_feedforward=transformer_dim_feedforward, num_encoder_layers=transformer_enc_layers,
            num_feature_levels=self.transformer_num_feature_levels)
        self.pe_layer = PositionEmbeddingSine(conv_dim // 2, normalize=True)
        self.mask_dim = mask_dim
        self.mask_features = Conv2d(conv_dim, mask_dim, kernel_size=1, stride=1, padding=0)
        self.oneformer_num_feature_levels = 3
        self.common_stride = common_stride
        stride = min(self.transformer_feature_strides)
        self.num_fpn_levels = int(np.log2(stride) - np.log2(self.common_stride))
        lateral_convs, output_convs = [], []
        use_bias = norm == ""
        for idx, in_channels in enumerate(self.feature_channels[:self.num_fpn_levels]):
            lateral_conv = Conv2d(in_channels, conv_dim, kernel_size=1, bias=use_bias, norm=get_norm(norm, conv_dim))
            output_conv = Conv2d(conv_dim, conv_dim, kernel_size=3, stride=1, padding=1, bias=use_bias,
                                 norm=get_norm(norm, conv_dim), activation=F.relu)
            self.add_module("adapter_{}".format(idx + 1), lateral_conv)
            self.add_module("layer_{}".format(idx + 1), output_conv)
            lateral_convs.append(lateral_conv)
            output_convs.append(output_conv)
        self.lateral_convs = lateral_convs[::-1]
        self.output_convs = output_convs[::-1]

    @classmethod
    def from_config(cls, cfg, input_shape: Dict[str, ShapeSpec]):
        return {
            "input_shape": {k: v for k, v in input_shape.items() if k in cfg.MODEL.SEM_SEG_HEAD.IN_FEATURES},
            "conv_dim": cfg.MODEL.SEM_SEG_HEAD.CONVS_DIM,
            "mask_dim": cfg.MODEL.SEM_SEG_HEAD.MASK_DIM,
            "norm": cfg.MODEL.SEM_SEG_HEAD.NORM,
            "transformer_dropout": cfg.MODEL.ONE_FORMER.DROPOUT,
            "transformer_nheads": cfg.MODEL.ONE_FORMER.NHEADS,
            "transformer_dim_feedforward": 1024,   # fixed for the deformable encoder (reference msdeformattn.py:326-328)
            "transformer_enc_layers": cfg.MODEL.SEM_SEG_HEAD.TRANSFORMER_ENC_LAYERS,
            "transformer_in_features": cfg.MODEL.SEM_SEG_HEAD.DEFORMABLE_TRANSFORMER_ENCODER_IN_FEATURES,
            "common_stride": cfg.MODEL.SEM_SEG_HEAD.COMMON_STRIDE,
        }

    def forward_features(self, features):
        srcs, pos, shapes = [], [], []
        for idx, f in enumerate(self.transformer_in_features[::-1]):
            x = features[f].float()
            B, _, H, W = x.shape
            shapes.append((H, W))
            srcs.append(_conv1x1_gn(_tokens(x), self.input_proj[idx][0], self.input_proj[idx][1], H, W))
            pos.append(self.pe_layer.tokens(B, H, W, x.device))
        y, spatial_shapes, level_start_index, _ = self.transformer(srcs, pos, shapes)
        bs = y.shape[0]
        out, start = [], 0
        for (H, W) in shapes:                          # channels-last views of the encoder output, no copies
            out.append(y[:, start:start + H * W].view(bs, H, W, -1).permute(0, 3, 1, 2))
            start += H * W
        last_tok = None
        for idx, f in enumerate(self.in_features[:self.num_fpn_levels][::-1]):
            x = features[f].float()
            B, _, H, W = x.shape
            lat, outc = self.lateral_convs[idx], self.output_convs[idx]
            fused = (isinstance(lat.norm, nn.GroupNorm) and isinstance(outc.norm, nn.GroupNorm) and _gn_tokens_ok(lat.norm)
                     and _gn_tokens_ok(outc.norm) and outc.kernel_size == (3, 3) and outc.bias is None and outc.stride == (1, 1)
                     and outc.padding == (1, 1) and outc.activation is F.relu and outc.in_channels % 8 == 0)
            if fused:
                # the whole branch on token matrices: GN + top-down merge in one pass (bf16 out = the conv's patch source),
                # conv as patch GEMM, GN + ReLU in one pass (bf16 out = the next GEMM's operand)
                prev = out[-1].permute(0, 2, 3, 1)                                          # (B, Hs, Ws, C) view
                if os.environ.get("UENC_FPN_CHAIN", "1") != "0":
                    # conv + GroupNorm as one autograd node each: the GroupNorm's input gradient stays bf16 on its way into the conv's GEMMs
                    yy = ops.linear_group_norm(_tokens(x), lat, lat.norm, add_src=prev, add_hw=(H, W), out_dtype=torch.bfloat16)
                    last_tok = ops.conv3x3_group_norm(yy.view(B, H, W, -1), outc.weight, outc.norm, relu=True, out_dtype=torch.bfloat16)
                else:                                                                       # (the two-node form, A/B)
                    yy = _conv1x1_gn(_tokens(x), lat, lat.norm, H, W, add_src=prev, add_hw=(H, W), out_dtype=torch.bfloat16)
                    z = ops.conv3x3(yy.view(B, H, W, -1), outc.weight)                      # (B, HW, C) fp32
                    last_tok = ops.group_norm_tokens(z, outc.norm, relu=True, out_dtype=torch.bfloat16)
                out.append(last_tok.view(B, H, W, -1).permute(0, 3, 1, 2))
                continue
            last_tok = None
            cur = _conv1x1_gn(_tokens(x), lat, lat.norm, H, W).transpose(1, 2).reshape(B, -1, H, W)
            yy = cur + F.interpolate(out[-1].float(), size=(H, W), mode="bilinear", align_corners=False)
            if outc.kernel_size == (3, 3) and outc.bias is None and outc.stride == (1, 1) and outc.padding == (1, 1):
                z = ops.conv3x3(yy.permute(0, 2, 3, 1), outc.weight).view(B, H, W, -1).permute(0, 3, 1, 2)   # im2col + MFMA GEMM
                if outc.norm is not None:
                    z = outc.norm(z)
                out.append(outc.activation(z) if outc.activation is not None else z)
            else:
                out.append(outc(yy))
        multi_scale_features = out[:self.oneformer_num_feature_levels]
        last = out[-1]
        B, _, H, W = last.shape
        mf = ops.linear(last_tok if last_tok is not None else _tokens(last), self.mask_features.weight, self.mask_features.bias,
                        out_dtype=torch.float32)
        # mask features stay channels-last in memory ((B, C, H, W) view): the decoder consumes them as tokens
        return mf.view(B, H, W, -1).permute(0, 3, 1, 2), out[0], multi_scale_features
