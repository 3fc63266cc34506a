"""CPU restatement (fp32, plain PyTorch) of the DiNAT backbone path.

TEST INFRASTRUCTURE, like oracle/torch_ref.py: only `tests/` and `__graft_entry__.smoke()` import it.

Parity status: **PARITY UNPINNED** for the neighbourhood attention itself.  The reference's `backbone/dinat.py`
imports `natten.NeighborhoodAttention2D` (`dinat.py:14`, used `:77-79, 94, 100`) from the wheel `natten==0.14.4`
(`requirements.txt:18-19`), which is neither vendored under /root/reference nor installed here, and the reference
holds no test, fixture or golden vector for it.  What is restated below is NATTEN 0.14.4's published algorithm:

  * `NeighborhoodAttention2D.forward`: zero-pad right / bottom up to kernel_size * dilation, `qkv` Linear,
    `q * head_dim**-0.5`, QK^T over each pixel's k x k neighbourhood + `rpb[nH, 2k-1, 2k-1]`, softmax over the k*k
    neighbours, attention x V, crop, `proj` Linear;
  * the neighbourhood of NATTEN's `get_window_start` / `get_pb_start`: with dilation d, pixel i = p * d + r only sees pixels of
    its own residue class r; inside the class (length L_r) the window is the k consecutive positions starting at
    clamp(p - k // 2, 0, L_r - k) -- clamped inside the image, never zero-padded -- and neighbour j of the window carries the
    bias entry (start + j - p) + k - 1 along that axis.

`na2d` is written twice -- as a gather over explicit neighbour indices and as dense attention under an explicit
neighbourhood mask -- and tests/test_dinat_cpu.py checks the two against each other and against scalar Python loops, so
the HIP kernels are at least pinned to a statement that is consistent three ways.  Everything around the attention
(ConvTokenizer, ConvDownsampler, NATLayer, per-stage norms: `dinat.py:17-45, 67-103, 139-227`) is the reference's own
code and is restated from it; citations are relative to /root/reference/model/modeling/backbone/.
"""
import math
from dataclasses import dataclass
from typing import Dict, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F

Tensor = torch.Tensor
SD = Dict[str, Tensor]


@dataclass
class DiNATCfg:  # config.py:217-237 (defaults: the "mini" variant), dinat.py:139-156
    embed_dim: int = 64
    mlp_ratio: float = 3.0
    depths: Sequence[int] = (3, 4, 18, 5)
    num_heads: Sequence[int] = (2, 4, 8, 16)
    kernel_size: int = 7
    dilations: Optional[Sequence[Sequence[int]]] = None
    out_indices: Sequence[int] = (0, 1, 2, 3)


# DiNAT-L hyper-parameters (not in the reference's config.py): embed 192, depths 3-4-18-5, heads 6-12-24-48,
# kernel 7, mlp_ratio 2.  Dilations depend on the input resolution and are passed by the caller.
def dinat_l(dilations=None) -> DiNATCfg:
    return DiNATCfg(192, 2.0, (3, 4, 18, 5), (6, 12, 24, 48), 7, dilations)


# ----------------------------------------------------------------------------
# neighbourhood geometry (NATTEN 0.14.4 get_window_start / get_pb_start, restated per residue class)
# ----------------------------------------------------------------------------
def axis_neighbours(length: int, k: int, d: int) -> Tuple[Tensor, Tensor]:
    """For every index i of an axis: the k neighbour indices (length, k) and their rpb indices (length, k) in [0, 2k-1)."""
    assert length >= k * d, "NATTEN pads the input to kernel_size * dilation first"
    i = torch.arange(length)
    r, p = i % d, i // d
    L = (length - r + d - 1) // d                      # members of the residue class
    start = torch.minimum(torch.clamp(p - k // 2, min=0), L - k)
    j = torch.arange(k)
    nb = (start[:, None] + j[None, :]) * d + r[:, None]
    pb = start[:, None] + j[None, :] - p[:, None] + (k - 1)
    return nb, pb


def na2d(q: Tensor, k: Tensor, v: Tensor, rpb: Optional[Tensor], ks: int, d: int) -> Tensor:
    """q (already scaled), k, v: (B, nH, H, W, hd); rpb (nH, 2ks-1, 2ks-1) -> (B, nH, H, W, hd).  Gather form."""
    B, nH, H, W, hd = q.shape
    ny, py = axis_neighbours(H, ks, d)
    nx, px = axis_neighbours(W, ks, d)
    kk = k[:, :, ny][:, :, :, :, nx]                     # (B, nH, H, ks, W, ks, hd)
    vv = v[:, :, ny][:, :, :, :, nx]
    s = torch.einsum("bhyxc,bhyixjc->bhyxij", q, kk)
    if rpb is not None:
        s = s + rpb[:, py][:, :, :, px].permute(0, 1, 3, 2, 4)          # (nH, H, ks, W, ks) -> (nH, H, W, ks, ks)
    a = s.reshape(B, nH, H, W, ks * ks).softmax(-1).reshape(B, nH, H, W, ks, ks)
    return torch.einsum("bhyxij,bhyixjc->bhyxc", a, vv)


def na2d_dense(q: Tensor, k: Tensor, v: Tensor, rpb: Optional[Tensor], ks: int, d: int) -> Tensor:
    """The same as ordinary softmax attention over all H*W keys under an explicit neighbourhood mask (small inputs only)."""
    B, nH, H, W, hd = q.shape
    ny, py = axis_neighbours(H, ks, d)
    nx, px = axis_neighbours(W, ks, d)
    bias = torch.full((nH, H, W, H, W), float("-inf"))
    for y in range(H):
        for x in range(W):
            for i in range(ks):
                for j in range(ks):
                    bias[:, y, x, ny[y, i], nx[x, j]] = rpb[:, py[y, i], px[x, j]] if rpb is not None else 0.0
    s = torch.einsum("bhyxc,bhuvc->bhyxuv", q, k) + bias
    a = s.reshape(B, nH, H, W, H * W).softmax(-1).reshape(B, nH, H, W, H, W)
    return torch.einsum("bhyxuv,bhuvc->bhyxc", a, v)


def _ln(x: Tensor, sd: SD, p: str) -> Tensor:
    return F.layer_norm(x, (x.shape[-1],), sd[p + ".weight"], sd[p + ".bias"], 1e-5)


def _lin(x: Tensor, sd: SD, p: str) -> Tensor:
    return F.linear(x, sd[p + ".weight"], sd.get(p + ".bias"))


def neighborhood_attention(x: Tensor, sd: SD, p: str, nH: int, ks: int, d: int) -> Tensor:
    """natten.NeighborhoodAttention2D.forward on (B, H, W, C); parameters p.qkv / p.rpb / p.proj."""
    B, Hp, Wp, C = x.shape
    ws = ks * d
    pad_r, pad_b = max(0, ws - Wp), max(0, ws - Hp)
    if pad_r or pad_b:
        x = F.pad(x, (0, 0, 0, pad_r, 0, pad_b))
    _, H, W, _ = x.shape
    hd = C // nH
    qkv = _lin(x, sd, p + ".qkv").reshape(B, H, W, 3, nH, hd).permute(3, 0, 4, 1, 2, 5)
    o = na2d(qkv[0] * hd ** -0.5, qkv[1], qkv[2], sd.get(p + ".rpb"), ks, d)
    o = o.permute(0, 2, 3, 1, 4).reshape(B, H, W, C)
    if pad_r or pad_b:
        o = o[:, :Hp, :Wp, :]
    return _lin(o, sd, p + ".proj")


def nat_layer(x: Tensor, sd: SD, p: str, nH: int, ks: int, d: int, branch_scale=None) -> Tensor:
    """NATLayer.forward without layer scale (dinat.py:90-97).  branch_scale = None: eval (DropPath identity); else two (B,)
    tensors of timm DropPath's per-sample multipliers for the attention and the MLP branch (training mode, dinat.py:95-96)."""
    a = neighborhood_attention(_ln(x, sd, p + ".norm1"), sd, p + ".attn", nH, ks, d)
    x = x + (a if branch_scale is None else a * branch_scale[0].view(-1, 1, 1, 1))
    h = _lin(F.gelu(_lin(_ln(x, sd, p + ".norm2"), sd, p + ".mlp.fc1")), sd, p + ".mlp.fc2")
    return x + (h if branch_scale is None else h * branch_scale[1].view(-1, 1, 1, 1))


def dinat_backbone(img: Tensor, sd: SD, cfg: DiNATCfg, prefix: str = "backbone.") -> Dict[str, Tensor]:
    """DiNAT.forward (dinat.py:207-227): img (B, 3, H, W) normalised -> {"res2".."res5"} NCHW."""
    # ConvTokenizer (dinat.py:17-33): two 3x3 stride-2 convolutions, NHWC, LayerNorm
    x = F.conv2d(img, sd[prefix + "patch_embed.proj.0.weight"], sd[prefix + "patch_embed.proj.0.bias"], stride=2, padding=1)
    x = F.conv2d(x, sd[prefix + "patch_embed.proj.1.weight"], sd[prefix + "patch_embed.proj.1.bias"], stride=2, padding=1)
    x = _ln(x.permute(0, 2, 3, 1), sd, prefix + "patch_embed.norm")
    outs = {}
    for i, depth in enumerate(cfg.depths):
        for j in range(depth):
            d = 1 if cfg.dilations is None else cfg.dilations[i][j]       # (an IndexError here is the reference's, dinat.py:120)
            x = nat_layer(x, sd, f"{prefix}levels.{i}.blocks.{j}", cfg.num_heads[i], cfg.kernel_size, d or 1)
        if i in cfg.out_indices:
            outs[f"res{i + 2}"] = _ln(x, sd, f"{prefix}norm{i}").permute(0, 3, 1, 2).contiguous()
        if i < len(cfg.depths) - 1:                                       # ConvDownsampler (dinat.py:36-45)
            x = F.conv2d(x.permute(0, 3, 1, 2), sd[f"{prefix}levels.{i}.downsample.reduction.weight"], None, stride=2, padding=1)
            x = _ln(x.permute(0, 2, 3, 1), sd, f"{prefix}levels.{i}.downsample.norm")
    return outs


def dinat_param_shapes(cfg: DiNATCfg, prefix: str = "backbone.") -> Dict[str, Tuple[int, ...]]:
    C0, ks = cfg.embed_dim, cfg.kernel_size
    s = {prefix + "patch_embed.proj.0.weight": (C0 // 2, 3, 3, 3), prefix + "patch_embed.proj.0.bias": (C0 // 2,),
         prefix + "patch_embed.proj.1.weight": (C0, C0 // 2, 3, 3), prefix + "patch_embed.proj.1.bias": (C0,),
         prefix + "patch_embed.norm.weight": (C0,), prefix + "patch_embed.norm.bias": (C0,)}
    for i, depth in enumerate(cfg.depths):
        C = C0 * 2 ** i
        hid = int(C * cfg.mlp_ratio)
        for j in range(depth):
            p = f"{prefix}levels.{i}.blocks.{j}"
            s.update({p + ".norm1.weight": (C,), p + ".norm1.bias": (C,), p + ".attn.rpb": (cfg.num_heads[i], 2 * ks - 1, 2 * ks - 1),
                      p + ".attn.qkv.weight": (3 * C, C), p + ".attn.qkv.bias": (3 * C,),
                      p + ".attn.proj.weight": (C, C), p + ".attn.proj.bias": (C,),
                      p + ".norm2.weight": (C,), p + ".norm2.bias": (C,),
                      p + ".mlp.fc1.weight": (hid, C), p + ".mlp.fc1.bias": (hid,),
                      p + ".mlp.fc2.weight": (C, hid), p + ".mlp.fc2.bias": (C,)})
        if i < len(cfg.depths) - 1:
            s.update({f"{prefix}levels.{i}.downsample.reduction.weight": (2 * C, C, 3, 3),
                      f"{prefix}levels.{i}.downsample.norm.weight": (2 * C,), f"{prefix}levels.{i}.downsample.norm.bias": (2 * C,)})
        if i in cfg.out_indices:
            s.update({f"{prefix}norm{i}.weight": (C,), f"{prefix}norm{i}.bias": (C,)})
    return s
