"""Swin Transformer backbone on HIP kernels — drop-in for the reference's `D2SwinTransformer`.

Mirrors the public surface of reference model/modeling/backbone/swin.py: same class names, same
constructor arguments, same parameter / buffer names and shapes (so reference checkpoints load:
`backbone.layers.{s}.blocks.{i}.attn.relative_position_bias_table`, `...relative_position_index`
as a persistent buffer, ...), same `forward(x) -> {"res2".."res5"}` contract, registered as
`D2SwinTransformer` in `BACKBONE_REGISTRY`.

What differs is how a block runs: one autograd Function per block (`ops.SwinBlockFn`) launching the
fused kernels — LayerNorm, bf16 MFMA GEMMs with bias/GELU/residual epilogues and the shifted-window
attention kernel that folds pad / roll / partition / reverse / crop into addressing — instead of the
reference's ~40 ATen calls and 4-6 full-tensor copies per block (swin.py:250-289).

Stochastic depth (DropPath) is applied in training mode as per-sample scales of the two residual
branches (`ops.drop_path_scales`; drawn from torch's CPU generator, so the random stream differs from the
reference's device-side draw).  Not reproduced (documented gaps): the MLP / attention dropout
layers (rate 0 in every shipped config: `DROP_RATE`, `ATTN_DROP_RATE` = 0, config.py:192-214); APE;
activation checkpointing.
"""
import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from ... import ops
from ...d2 import BACKBONE_REGISTRY, Backbone, ShapeSpec


def to_2tuple(x):
    return tuple(x) if isinstance(x, (tuple, list)) else (x, x)


class Mlp(nn.Module):
    """fc1 -> GELU -> fc2 (swin.py:21-41); parameters only, the block Function runs it."""

    def __init__(self, in_features, hidden_features=None, out_features=None, act_layer=nn.GELU, drop=0.0):
        super().__init__()
        out_features = out_features or in_features
        hidden_features = hidden_features or in_features
        self.fc1 = nn.Linear(in_features, hidden_features)
        self.fc2 = nn.Linear(hidden_features, out_features)

    def forward(self, x):
        return ops.mlp(x, [self.fc1.weight, self.fc1.bias, self.fc2.weight, self.fc2.bias], act="gelu")


class WindowAttention(nn.Module):
    """Parameter container of W-MSA / SW-MSA (swin.py:74-129)."""

    def __init__(self, dim, window_size, num_heads, qkv_bias=True, qk_scale=None, attn_drop=0.0, proj_drop=0.0):
        super().__init__()
        self.dim, self.window_size, self.num_heads = dim, window_size, num_heads
        head_dim = dim // num_heads
        if head_dim != 32:
            raise ValueError(f"the HIP window-attention kernel is built for head_dim 32 (got {head_dim}); every "
                             "Swin-T/S/B/L stage satisfies this")
        self.scale = qk_scale or head_dim ** -0.5
        ws = window_size[0]
        self.relative_position_bias_table = nn.Parameter(torch.zeros((2 * ws - 1) * (2 * ws - 1), num_heads))
        ys, xs = torch.meshgrid(torch.arange(ws), torch.arange(ws), indexing="ij")
        ys, xs = ys.reshape(-1), xs.reshape(-1)
        idx = (ys[:, None] - ys[None, :] + ws - 1) * (2 * ws - 1) + (xs[:, None] - xs[None, :] + ws - 1)
        self.register_buffer("relative_position_index", idx)       # kept for state-dict parity; kernels index arithmetically
        self.qkv = nn.Linear(dim, dim * 3, bias=qkv_bias)
        self.proj = nn.Linear(dim, dim)
        nn.init.trunc_normal_(self.relative_position_bias_table, std=0.02)


class SwinTransformerBlock(nn.Module):
    def __init__(self, dim, num_heads, window_size=7, shift_size=0, mlp_ratio=4.0, qkv_bias=True, qk_scale=None,
                 drop=0.0, attn_drop=0.0, drop_path=0.0, act_layer=nn.GELU, norm_layer=nn.LayerNorm):
        super().__init__()
        assert 0 <= shift_size < window_size, "shift_size must in 0-window_size"
        self.dim, self.num_heads, self.window_size, self.shift_size, self.mlp_ratio = dim, num_heads, window_size, shift_size, mlp_ratio
        self.norm1 = norm_layer(dim)
        self.attn = WindowAttention(dim, to_2tuple(window_size), num_heads, qkv_bias, qk_scale, attn_drop, drop)
        self.drop_path_rate = drop_path
        self.norm2 = norm_layer(dim)
        self.mlp = Mlp(dim, int(dim * mlp_ratio), act_layer=act_layer, drop=drop)
        self.H = self.W = None

    def _params(self):
        a, m = self.attn, self.mlp
        qb = a.qkv.bias if a.qkv.bias is not None else torch.zeros(3 * self.dim, device=a.qkv.weight.device)
        return [self.norm1.weight, self.norm1.bias, a.qkv.weight, qb, a.relative_position_bias_table,
                a.proj.weight, a.proj.bias, self.norm2.weight, self.norm2.bias,
                m.fc1.weight, m.fc1.bias, m.fc2.weight, m.fc2.bias]

    def forward(self, x, mask_matrix=None):
        """x (B, H*W, C) fp32.  `mask_matrix` is accepted for signature parity and ignored: the kernel
        derives the 9-region shift mask (-100 off-region) from coordinates (swin.py:414-440)."""
        B, L, C = x.shape
        H, W = self.H, self.W
        assert L == H * W, "input feature has wrong size"
        dp = None
        if self.training and self.drop_path_rate > 0.0:      # stochastic depth: one draw per residual branch and sample (swin.py:279, 289)
            dp = (ops.drop_path_scales(B, self.drop_path_rate), ops.drop_path_scales(B, self.drop_path_rate))
        return ops.swin_block(x.float(), H, W, self.window_size, self.shift_size, self.num_heads, self.attn.scale,
                              self._params(), dp)


class PatchMerging(nn.Module):
    """2x2 neighbourhood concat (order (0,0),(1,0),(0,1),(1,1)) -> LN(4C) -> Linear 4C->2C (swin.py:298-337)."""

    def __init__(self, dim, norm_layer=nn.LayerNorm):
        super().__init__()
        self.dim = dim
        self.reduction = nn.Linear(4 * dim, 2 * dim, bias=False)
        self.norm = norm_layer(4 * dim)

    def forward(self, x, H, W):
        B, L, C = x.shape
        assert L == H * W, "input feature has wrong size"
        x = x.view(B, H, W, C)
        if x.dtype == torch.float32 and 4 * C <= 6144 and C % 4 == 0 and not ops.is_exact():
            # pad-to-even, the four strided slices, the concat and the LayerNorm are one kernel (index arithmetic)
            xn = ops.patch_merge_ln(x, self.norm.weight, self.norm.bias, self.norm.eps)
        else:
            if H % 2 == 1 or W % 2 == 1:
                x = F.pad(x, (0, 0, 0, W % 2, 0, H % 2))
            x = torch.cat([x[:, 0::2, 0::2], x[:, 1::2, 0::2], x[:, 0::2, 1::2], x[:, 1::2, 1::2]], -1)
            x = x.reshape(B, -1, 4 * C)
            xn = ops.layer_norm(x, self.norm.weight, self.norm.bias, out_dtype=torch.bfloat16)
        return ops.linear(xn, self.reduction.weight, None, out_dtype=torch.float32)


class BasicLayer(nn.Module):
    def __init__(self, dim, depth, num_heads, window_size=7, mlp_ratio=4.0, qkv_bias=True, qk_scale=None, drop=0.0,
                 attn_drop=0.0, drop_path=0.0, norm_layer=nn.LayerNorm, downsample=None, use_checkpoint=False):
        super().__init__()
        self.window_size, self.shift_size, self.depth = window_size, window_size // 2, depth
        self.blocks = nn.ModuleList([
            SwinTransformerBlock(dim, num_heads, window_size, 0 if i % 2 == 0 else window_size // 2, mlp_ratio, qkv_bias,
                                 qk_scale, drop, attn_drop, drop_path[i] if isinstance(drop_path, list) else drop_path,
                                 norm_layer=norm_layer) for i in range(depth)])
        self.downsample = downsample(dim=dim, norm_layer=norm_layer) if downsample is not None else None

    def forward(self, x, H, W):
        for blk in self.blocks:
            blk.H, blk.W = H, W
            x = blk(x, None)
        if self.downsample is not None:
            x_down = self.downsample(x, H, W)
            return x, H, W, x_down, (H + 1) // 2, (W + 1) // 2
        return x, H, W, x, H, W


class PatchEmbed(nn.Module):
    """4x4 stride-4 conv as a K=48 GEMM over patch rows, then LayerNorm (swin.py:456-495)."""

    def __init__(self, patch_size=4, in_chans=3, embed_dim=96, norm_layer=None):
        super().__init__()
        self.patch_size = to_2tuple(patch_size)
        self.in_chans, self.embed_dim = in_chans, embed_dim
        self.proj = nn.Conv2d(in_chans, embed_dim, kernel_size=self.patch_size, stride=self.patch_size)
        self.norm = norm_layer(embed_dim) if norm_layer is not None else None

    def forward_tokens(self, x):
        """(B, 3, H, W) -> tokens (B, L, C) fp32, Wh, Ww."""
        ph, pw = self.patch_size
        _, _, H, W = x.shape
        if W % pw != 0:
            x = F.pad(x, (0, pw - W % pw))
        if H % ph != 0:
            x = F.pad(x, (0, 0, 0, ph - H % ph))
        B, Cin, H, W = x.shape
        Wh, Ww = H // ph, W // pw
        patches = x.view(B, Cin, Wh, ph, Ww, pw).permute(0, 2, 4, 1, 3, 5).reshape(B, Wh * Ww, Cin * ph * pw)
        if self.norm is not None:
            y = ops.linear(patches, self.proj.weight, self.proj.bias, out_dtype=torch.float32)
            y = ops.layer_norm(y, self.norm.weight, self.norm.bias, out_dtype=torch.float32)
        else:
            y = ops.linear(patches, self.proj.weight, self.proj.bias, out_dtype=torch.float32)
        return y, Wh, Ww

    def forward(self, x):
        y, Wh, Ww = self.forward_tokens(x)
        return y.transpose(1, 2).reshape(-1, self.embed_dim, Wh, Ww)


class SwinTransformer(nn.Module):
    def __init__(self, pretrain_img_size=224, patch_size=4, in_chans=3, embed_dim=96, depths=[2, 2, 6, 2],
                 num_heads=[3, 6, 12, 24], window_size=7, mlp_ratio=4.0, qkv_bias=True, qk_scale=None, drop_rate=0.0,
                 attn_drop_rate=0.0, drop_path_rate=0.2, norm_layer=nn.LayerNorm, ape=False, patch_norm=True,
                 out_indices=(0, 1, 2, 3), frozen_stages=-1, use_checkpoint=False):
        super().__init__()
        # MODEL.SWIN.DROP_RATE / ATTN_DROP_RATE (pos_drop, Mlp.drop, proj_drop, attn_drop of reference swin.py:33-40, 124-126, 580): 0.0 in
        # every shipped config.  Identity in eval mode, so such a model builds and infers; a TRAINING forward with them refuses
        # (forward below) rather than silently train differently from the reference.  Stochastic depth (DROP_PATH_RATE) is implemented.
        self.drop_rate, self.attn_drop_rate = float(drop_rate), float(attn_drop_rate)
        self.pretrain_img_size, self.num_layers, self.embed_dim = pretrain_img_size, len(depths), embed_dim
        self.ape, self.patch_norm, self.out_indices, self.frozen_stages = ape, patch_norm, out_indices, frozen_stages
        self.patch_embed = PatchEmbed(patch_size, in_chans, embed_dim, norm_layer if patch_norm else None)
        if ape:          # absolute position embedding at the pre-training resolution (reference swin.py:566-578), resized per input below
            pi, ps = to_2tuple(pretrain_img_size), to_2tuple(patch_size)
            self.absolute_pos_embed = nn.Parameter(torch.zeros(1, embed_dim, pi[0] // ps[0], pi[1] // ps[1]))
            nn.init.trunc_normal_(self.absolute_pos_embed, std=0.02)
        dpr = [x.item() for x in torch.linspace(0, drop_path_rate, sum(depths))]
        self.layers = nn.ModuleList()
        for i in range(self.num_layers):
            self.layers.append(BasicLayer(
                dim=int(embed_dim * 2 ** i), depth=depths[i], num_heads=num_heads[i], window_size=window_size,
                mlp_ratio=mlp_ratio, qkv_bias=qkv_bias, qk_scale=qk_scale, drop=drop_rate, attn_drop=attn_drop_rate,
                drop_path=dpr[sum(depths[:i]):sum(depths[:i + 1])], norm_layer=norm_layer,
                downsample=PatchMerging if i < self.num_layers - 1 else None, use_checkpoint=use_checkpoint))
        self.num_features = [int(embed_dim * 2 ** i) for i in range(self.num_layers)]
        for i in out_indices:
            self.add_module(f"norm{i}", norm_layer(self.num_features[i]))
        self._freeze_stages()

    def _freeze_stages(self):
        if self.frozen_stages >= 0:
            self.patch_embed.eval()
            for p in self.patch_embed.parameters():
                p.requires_grad = False
        if self.frozen_stages >= 2:
            for i in range(0, self.frozen_stages - 1):
                m = self.layers[i]
                m.eval()
                for p in m.parameters():
                    p.requires_grad = False

    def init_weights(self, pretrained=None):
        """No-op, as in the reference (swin.py:635-649 defines but never applies its initialiser)."""

    def forward(self, x):
        if self.training and (self.drop_rate != 0.0 or self.attn_drop_rate != 0.0):
            raise NotImplementedError("training with MODEL.SWIN.DROP_RATE / ATTN_DROP_RATE > 0 is not implemented in the fused Swin block "
                                      "(0.0 in every shipped config, config.py:192-214); inference with such a model is")
        x, Wh, Ww = self.patch_embed.forward_tokens(x)
        if self.ape:     # reference swin.py:656-661: bicubic resize of the embedding to the token grid, added before the first stage
            pe = F.interpolate(self.absolute_pos_embed, size=(Wh, Ww), mode="bicubic")
            x = x + pe.flatten(2).transpose(1, 2)
        outs = {}
        for i in range(self.num_layers):
            x_out, H, W, x, Wh, Ww = self.layers[i](x, Wh, Ww)
            if i in self.out_indices:
                n = getattr(self, f"norm{i}")
                o = ops.layer_norm(x_out, n.weight, n.bias, out_dtype=torch.float32)
                # (B, C, H, W)-shaped, stored channels-last: the 1x1 convs downstream read token rows directly
                outs[f"res{i + 2}"] = o.view(-1, H, W, self.num_features[i]).permute(0, 3, 1, 2)
        return outs

    def train(self, mode=True):
        """Keeps frozen stages frozen; returns None like the reference (swin.py:680-683)."""
        super().train(mode)
        self._freeze_stages()


@BACKBONE_REGISTRY.register()
class D2SwinTransformer(SwinTransformer, Backbone):
    def __init__(self, cfg, input_shape):
        s = cfg.MODEL.SWIN
        super().__init__(s.PRETRAIN_IMG_SIZE, s.PATCH_SIZE, 3, s.EMBED_DIM, s.DEPTHS, s.NUM_HEADS, s.WINDOW_SIZE,
                         s.MLP_RATIO, s.QKV_BIAS, s.QK_SCALE, s.DROP_RATE, s.ATTN_DROP_RATE, s.DROP_PATH_RATE,
                         nn.LayerNorm, s.APE, s.PATCH_NORM, use_checkpoint=s.USE_CHECKPOINT)
        self._out_features = s.OUT_FEATURES
        self._out_feature_strides = {"res2": 4, "res3": 8, "res4": 16, "res5": 32}
        self._out_feature_channels = {f"res{i + 2}": self.num_features[i] for i in range(4)}

    def forward(self, x):
        assert x.dim() == 4, f"SwinTransformer takes an input of shape (N, C, H, W). Got {x.shape} instead!"
        y = super().forward(x)
        return {k: v for k, v in y.items() if k in self._out_features}

    def output_shape(self):
        return {name: ShapeSpec(channels=self._out_feature_channels[name], stride=self._out_feature_strides[name])
                for name in self._out_features}

    @property
    def size_divisibility(self):
        return 32
