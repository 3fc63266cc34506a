"""DiNAT backbone on HIP kernels — drop-in for the reference's `D2DiNAT`.

Mirrors the public surface of reference model/modeling/backbone/dinat.py: same class names, constructor arguments,
parameter names and shapes (`backbone.patch_embed.proj.{0,1}`, `backbone.levels.{i}.blocks.{j}.attn.{qkv,rpb,proj}`,
`...downsample.reduction`, `backbone.norm{i}`), the same `forward(x) -> {"res2".."res5"}` contract, registered as
`D2DiNAT` in `BACKBONE_REGISTRY` with `cfg.MODEL.DiNAT.*` (config.add_dinat_config).

`NeighborhoodAttention2D` stands in for `natten.NeighborhoodAttention2D` (`dinat.py:14`): the reference takes it from
the un-vendored wheel natten==0.14.4, so its arithmetic is restated from NATTEN's published algorithm (oracle/
dinat_ref.py, parity unpinned) and runs in `csrc/na2d.hip`.  A NATLayer is one autograd Function (`ops.NATLayerFn`):
LayerNorm, bf16 MFMA GEMMs with bias / GELU / residual epilogues and the neighbourhood-attention kernels, on an
fp32 channels-last residual stream; the 3x3 stride-2 convolutions of the tokenizer and the downsamplers are patch
gathers + the same GEMMs (`ops.ConvS2Fn`).

Stochastic depth is applied in training mode as in the Swin path (`ops.drop_path_scales`).  Not reproduced: the
dropout layers (rate 0 in the shipped config, config.py:235-236); `layer_scale` (never passed by `D2DiNAT`,
`dinat.py:246-255`) raises.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from ... import ops
from ...d2 import BACKBONE_REGISTRY, Backbone, ShapeSpec


class ConvTokenizer(nn.Module):
    """Two 3x3 stride-2 convolutions + LayerNorm, NCHW image -> channels-last tokens at 1/4 resolution (dinat.py:17-33)."""

    def __init__(self, in_chans=3, embed_dim=96, norm_layer=None):
        super().__init__()
        self.proj = nn.Sequential(
            nn.Conv2d(in_chans, embed_dim // 2, kernel_size=(3, 3), stride=(2, 2), padding=(1, 1)),
            nn.Conv2d(embed_dim // 2, embed_dim, kernel_size=(3, 3), stride=(2, 2), padding=(1, 1)))
        self.norm = norm_layer(embed_dim) if norm_layer is not None else None

    def forward(self, x):
        x = ops.conv3x3_s2(x.permute(0, 2, 3, 1), self.proj[0].weight, self.proj[0].bias)
        x = ops.conv3x3_s2(x, self.proj[1].weight, self.proj[1].bias)
        if self.norm is not None:
            x = ops.layer_norm(x, self.norm.weight, self.norm.bias, out_dtype=torch.float32)
        return x


class ConvDownsampler(nn.Module):
    """3x3 stride-2 convolution C -> 2C without bias + LayerNorm on channels-last maps (dinat.py:36-45)."""

    def __init__(self, dim, norm_layer=nn.LayerNorm):
        super().__init__()
        self.reduction = nn.Conv2d(dim, 2 * dim, kernel_size=(3, 3), stride=(2, 2), padding=(1, 1), bias=False)
        self.norm = norm_layer(2 * dim)

    def forward(self, x):
        x = ops.conv3x3_s2(x, self.reduction.weight, None)
        return ops.layer_norm(x, self.norm.weight, self.norm.bias, out_dtype=torch.float32)


class Mlp(nn.Module):
    """fc1 -> GELU -> fc2 (dinat.py:48-64); parameters only, the layer Function runs it."""

    def __init__(self, in_features, hidden_features=None, out_features=None, act_layer=nn.GELU, drop=0.0):
        super().__init__()
        out_features = out_features or in_features
        hidden_features = hidden_features or in_features
        self.fc1 = nn.Linear(in_features, hidden_features)
        self.fc2 = nn.Linear(hidden_features, out_features)

    def forward(self, x):
        return ops.mlp(x, [self.fc1.weight, self.fc1.bias, self.fc2.weight, self.fc2.bias], act="gelu")


class NeighborhoodAttention2D(nn.Module):
    """natten.NeighborhoodAttention2D (natten==0.14.4): same constructor arguments, parameters `qkv`, `rpb`, `proj`."""

    def __init__(self, dim, num_heads, kernel_size, dilation=1, bias=True, qkv_bias=True, qk_scale=None, attn_drop=0.0, proj_drop=0.0):
        super().__init__()
        self.num_heads = num_heads
        self.head_dim = dim // num_heads
        if self.head_dim != 32:
            raise ValueError(f"the HIP neighbourhood-attention kernels are built for head_dim 32 (got {self.head_dim}); every DiNAT "
                             "variant satisfies this")
        self.scale = qk_scale or self.head_dim ** -0.5
        assert kernel_size > 1 and kernel_size % 2 == 1, f"Kernel size must be an odd number greater than 1, got {kernel_size}."
        assert kernel_size <= 13, "the HIP kernels are instantiated for kernel sizes 3..13"
        self.kernel_size = kernel_size
        self.dilation = dilation or 1
        assert self.dilation >= 1, f"Dilation must be greater than or equal to 1, got {dilation}."
        self.window_size = self.kernel_size * self.dilation
        self.qkv = nn.Linear(dim, dim * 3, bias=qkv_bias)
        if bias:
            self.rpb = nn.Parameter(torch.zeros(num_heads, 2 * kernel_size - 1, 2 * kernel_size - 1))
            nn.init.trunc_normal_(self.rpb, std=0.02, mean=0.0, a=-2.0, b=2.0)
        else:
            self.register_parameter("rpb", None)
        self.proj = nn.Linear(dim, dim)

    def forward(self, x, residual=None):
        """x (B, Hp, Wp, C) channels-last -> proj(NA(qkv(x))) [+ residual, fp32]; inputs smaller than the window are zero-padded
        right / bottom before qkv and cropped after the attention, as NATTEN does."""
        B, Hp, Wp, C = x.shape
        pad_r, pad_b = max(0, self.window_size - Wp), max(0, self.window_size - Hp)
        if pad_r or pad_b:
            x = F.pad(x, (0, 0, 0, pad_r, 0, pad_b))
        qkv = ops.linear(x, self.qkv.weight, self.qkv.bias)
        o = ops.na2d(qkv, self.rpb, self.num_heads, self.kernel_size, self.dilation, self.scale)
        if pad_r or pad_b:
            o = o[:, :Hp, :Wp, :]
        return ops.linear(o, self.proj.weight, self.proj.bias, residual=residual)


class NATLayer(nn.Module):
    def __init__(self, dim, num_heads, kernel_size=7, dilation=None, mlp_ratio=4.0, qkv_bias=True, qk_scale=None, drop=0.0,
                 attn_drop=0.0, drop_path=0.0, act_layer=nn.GELU, norm_layer=nn.LayerNorm, layer_scale=None):
        super().__init__()
        if layer_scale is not None:
            raise NotImplementedError("layer_scale is never set on the D2DiNAT path (dinat.py:246-255)")
        self.dim, self.num_heads, self.mlp_ratio = dim, num_heads, mlp_ratio
        self.norm1 = norm_layer(dim)
        self.attn = NeighborhoodAttention2D(dim, kernel_size=kernel_size, dilation=dilation, num_heads=num_heads, qkv_bias=qkv_bias,
                                            qk_scale=qk_scale, attn_drop=attn_drop, proj_drop=drop)
        self.drop_path_rate = drop_path
        self.norm2 = norm_layer(dim)
        self.mlp = Mlp(in_features=dim, hidden_features=int(dim * mlp_ratio), act_layer=act_layer, drop=drop)
        self.layer_scale = False

    def forward(self, x):
        """x (B, H, W, C) fp32 residual stream."""
        a = self.attn
        B, H, W, _ = x.shape
        dp = None
        if self.training and self.drop_path_rate > 0.0:      # stochastic depth: one draw per residual branch and sample (dinat.py:95-96)
            dp = (ops.drop_path_scales(B, self.drop_path_rate), ops.drop_path_scales(B, self.drop_path_rate))
        if H >= a.window_size and W >= a.window_size and a.rpb is not None and a.qkv.bias is not None:
            return ops.nat_layer(x, a.num_heads, a.kernel_size, a.dilation, a.scale,
                                 [self.norm1.weight, self.norm1.bias, a.qkv.weight, a.qkv.bias, a.rpb, a.proj.weight, a.proj.bias,
                                  self.norm2.weight, self.norm2.bias, self.mlp.fc1.weight, self.mlp.fc1.bias, self.mlp.fc2.weight,
                                  self.mlp.fc2.bias], dp)
        # small maps (NATTEN's padding path) and bias-free variants: the same kernels, composed op by op
        if dp is None:
            x = a(ops.layer_norm(x, self.norm1.weight, self.norm1.bias, out_dtype=torch.bfloat16), residual=x)
            h = ops.layer_norm(x, self.norm2.weight, self.norm2.bias, out_dtype=torch.bfloat16)
            return ops.mlp(h, [self.mlp.fc1.weight, self.mlp.fc1.bias, self.mlp.fc2.weight, self.mlp.fc2.bias], act="gelu", residual=x)
        s1, s2 = (torch.tensor(v, device=x.device).view(B, 1, 1, 1) for v in dp)
        x = x + s1 * a(ops.layer_norm(x, self.norm1.weight, self.norm1.bias, out_dtype=torch.bfloat16)).float()
        h = ops.layer_norm(x, self.norm2.weight, self.norm2.bias, out_dtype=torch.bfloat16)
        return x + s2 * ops.mlp(h, [self.mlp.fc1.weight, self.mlp.fc1.bias, self.mlp.fc2.weight, self.mlp.fc2.bias], act="gelu").float()


class NATBlock(nn.Module):
    def __init__(self, dim, depth, num_heads, kernel_size, dilations=None, downsample=True, mlp_ratio=4.0, qkv_bias=True, qk_scale=None,
                 drop=0.0, attn_drop=0.0, drop_path=0.0, norm_layer=nn.LayerNorm, layer_scale=None):
        super().__init__()
        self.dim, self.depth = dim, depth
        self.blocks = nn.ModuleList([
            NATLayer(dim=dim, num_heads=num_heads, kernel_size=kernel_size, dilation=None if dilations is None else dilations[i],
                     mlp_ratio=mlp_ratio, qkv_bias=qkv_bias, qk_scale=qk_scale, drop=drop, attn_drop=attn_drop,
                     drop_path=drop_path[i] if isinstance(drop_path, list) else drop_path, norm_layer=norm_layer, layer_scale=layer_scale)
            for i in range(depth)])
        self.downsample = None if not downsample else ConvDownsampler(dim=dim, norm_layer=norm_layer)

    def forward(self, x):
        for blk in self.blocks:
            x = blk(x)
        if self.downsample is None:
            return x, x
        return self.downsample(x), x


class DiNAT(nn.Module):
    def __init__(self, embed_dim, mlp_ratio, depths, num_heads, drop_path_rate=0.2, in_chans=3, kernel_size=7, dilations=None,
                 out_indices=(0, 1, 2, 3), qkv_bias=True, qk_scale=None, drop_rate=0.0, attn_drop_rate=0.0, norm_layer=nn.LayerNorm,
                 frozen_stages=-1, layer_scale=None, **kwargs):
        super().__init__()
        self.num_levels = len(depths)
        self.embed_dim = embed_dim
        self.num_features = [int(embed_dim * 2 ** i) for i in range(self.num_levels)]
        self.mlp_ratio = mlp_ratio
        self.patch_embed = ConvTokenizer(in_chans=in_chans, embed_dim=embed_dim, norm_layer=norm_layer)
        dpr = [x.item() for x in torch.linspace(0, drop_path_rate, sum(depths))]
        self.levels = nn.ModuleList()
        for i in range(self.num_levels):
            self.levels.append(NATBlock(
                dim=int(embed_dim * 2 ** i), depth=depths[i], num_heads=num_heads[i], kernel_size=kernel_size,
                dilations=None if dilations is None else dilations[i], mlp_ratio=self.mlp_ratio, qkv_bias=qkv_bias, qk_scale=qk_scale,
                drop=drop_rate, attn_drop=attn_drop_rate, drop_path=dpr[sum(depths[:i]):sum(depths[:i + 1])], norm_layer=norm_layer,
                downsample=(i < self.num_levels - 1), layer_scale=layer_scale))
        self.out_indices = out_indices
        for i_layer in self.out_indices:
            self.add_module(f"norm{i_layer}", norm_layer(self.num_features[i_layer]))
        self.frozen_stages = frozen_stages

    def _freeze_stages(self):
        if self.frozen_stages >= 0:
            self.patch_embed.eval()
            for param in self.patch_embed.parameters():
                param.requires_grad = False
        if self.frozen_stages >= 2:
            for i in range(0, self.frozen_stages - 1):
                m = self.network[i]          # (AttributeError as in the reference, dinat.py:202: there is no `self.network`)
                m.eval()
                for param in m.parameters():
                    param.requires_grad = False

    def train(self, mode=True):
        super().train(mode)
        self._freeze_stages()

    def forward_embeddings(self, x):
        return self.patch_embed(x)

    def forward_tokens(self, x):
        outs = {}
        for idx, level in enumerate(self.levels):
            x, xo = level(x)
            if idx in self.out_indices:
                n = getattr(self, f"norm{idx}")
                # (B, C, H, W)-shaped, stored channels-last: the 1x1 convs downstream read token rows directly
                outs[f"res{idx + 2}"] = ops.layer_norm(xo, n.weight, n.bias, out_dtype=torch.float32).permute(0, 3, 1, 2)
        return outs

    def forward(self, x):
        return self.forward_tokens(self.forward_embeddings(x))


@BACKBONE_REGISTRY.register()
class D2DiNAT(DiNAT, Backbone):
    def __init__(self, cfg, input_shape):
        c = cfg.MODEL.DiNAT
        super().__init__(embed_dim=c.EMBED_DIM, mlp_ratio=c.MLP_RATIO, depths=c.DEPTHS, num_heads=c.NUM_HEADS,
                         drop_path_rate=c.DROP_PATH_RATE, kernel_size=c.KERNEL_SIZE, out_indices=c.OUT_INDICES, dilations=c.DILATIONS)
        self._out_features = c.OUT_FEATURES
        self._out_feature_strides = {"res2": 4, "res3": 8, "res4": 16, "res5": 32}
        self._out_feature_channels = {f"res{i + 2}": self.num_features[i] for i in range(4)}

    def forward(self, x):
        assert x.dim() == 4, f"DiNAT takes an input of shape (N, C, H, W). Got {x.shape} instead!"
        y = super().forward(x)
        return {k: v for k, v in y.items() if k in self._out_features}

    def output_shape(self):
        return {name: ShapeSpec(channels=self._out_feature_channels[name], stride=self._out_feature_strides[name])
                for name in self._out_features}

    @property
    def size_divisibility(self):
        return 32
