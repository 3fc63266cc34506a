"""CPU restatement (fp32, plain PyTorch) of the reference's unified-encoder hot path.

TEST INFRASTRUCTURE.  This file is the *oracle*: only `tests/`, `__graft_entry__.smoke()`
and the `cpu_baseline` leg of `bench.py` may import it.  The shipped product
(`uni-encoder-code_amd/`) never does; it fails loudly when its HIP library is missing.

Parity status: PINNED.  Every function below is checked in the build container
against the reference's own modules imported from /root/reference
(`oracle/ref_loader.py`, `tests/test_oracle_vs_reference.py`) and against the
fixtures those modules produced (`tests/golden/*.npz`, made by
`oracle/make_golden.py`).  The reference holds no tests or golden vectors of its
own.

The restatement is functional: it takes a flat state dict with the reference's
parameter names and never builds nn.Modules.  Window attention
is written as an index gather/scatter (the form the HIP kernel uses) rather than
the reference's pad -> roll -> partition copies, so it independently pins the
shift / pad / mask addressing.  Citations are relative to /root/reference/model.
"""
import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F

Tensor = torch.Tensor
SD = Dict[str, Tensor]


# ----------------------------------------------------------------------------
# configuration
# ----------------------------------------------------------------------------
@dataclass
class SwinCfg:  # modeling/backbone/swin.py:526-547, config.py:192-214
    embed_dim: int = 96
    depths: Sequence[int] = (2, 2, 6, 2)
    num_heads: Sequence[int] = (3, 6, 12, 24)
    window_size: int = 7
    mlp_ratio: float = 4.0
    patch_size: int = 4
    qk_scale: Optional[float] = None
    out_features: Sequence[str] = ("res2", "res3", "res4", "res5")


@dataclass
class HeadCfg:  # configs/cityscapes/oneformer_R50_bs16_90k.yaml
    conv_dim: int = 256
    mask_dim: int = 256
    hidden_dim: int = 256
    nheads: int = 8
    enc_layers: int = 6          # SEM_SEG_HEAD.TRANSFORMER_ENC_LAYERS
    enc_ffn: int = 1024          # hard-coded, pixel_decoder/msdeformattn.py:328
    n_points: int = 4
    dec_layers: int = 9          # ONE_FORMER.DEC_LAYERS - 1
    class_dec_layers: int = 2
    dim_feedforward: int = 2048
    num_queries: int = 150
    num_classes: int = 19
    use_task_norm: bool = True
    transformer_in_features: Sequence[str] = ("res3", "res4", "res5")


@dataclass
class ModelCfg:
    swin: SwinCfg = field(default_factory=SwinCfg)
    head: HeadCfg = field(default_factory=HeadCfg)
    pixel_mean: Sequence[float] = (123.675, 116.280, 103.530)
    pixel_std: Sequence[float] = (58.395, 57.120, 57.375)
    size_divisibility: int = 32
    task_seq_len: int = 77


SWIN_T = SwinCfg(96, (2, 2, 6, 2), (3, 6, 12, 24), 7)
SWIN_L = SwinCfg(192, (2, 2, 18, 2), (6, 12, 24, 48), 12)

# CLIP-BPE ids of the three task prompts (data/tokenizer.py:87-117 applied to
# "The task is {panoptic,semantic,instance}"), SURVEY.md §8c.  Zero-padded to 77.
TASK_TOKEN_IDS = {
    "The task is panoptic": [49406, 518, 10549, 533, 1072, 24755, 49407],
    "The task is semantic": [49406, 518, 10549, 533, 29119, 1550, 49407],
    "The task is instance": [49406, 518, 10549, 533, 34572, 49407],
}


def task_tokens(task: str, seq_len: int = 77) -> Tensor:
    ids = TASK_TOKEN_IDS[task]
    out = torch.zeros(seq_len, dtype=torch.long)
    out[: len(ids)] = torch.tensor(ids)
    return out


# ----------------------------------------------------------------------------
# small helpers
# ----------------------------------------------------------------------------
def _ln(x: Tensor, sd: SD, p: str, eps: float = 1e-5) -> Tensor:
    return F.layer_norm(x, (x.shape[-1],), sd[p + ".weight"], sd[p + ".bias"], eps)


def _lin(x: Tensor, sd: SD, p: str) -> Tensor:
    return F.linear(x, sd[p + ".weight"], sd.get(p + ".bias"))


def _gn(x: Tensor, sd: SD, p: str, groups: int = 32) -> Tensor:
    return F.group_norm(x, groups, sd[p + ".weight"], sd[p + ".bias"], 1e-5)


# ----------------------------------------------------------------------------
# A1/A2: shifted-window attention as gather / scatter
# ----------------------------------------------------------------------------
def window_layout(H: int, W: int, ws: int, shift: int) -> Tuple[Tensor, Tensor]:
    """Token index and mask-region id of every slot of every window.

    Restates backbone/swin.py:250-271 (zero pad to a multiple of ws, cyclic roll by
    -shift, partition) and :414-440 (9-region shift mask) as pure index arithmetic.
    Slot (i, j) of the padded+rolled grid holds original token ((i+shift) % Hp,
    (j+shift) % Wp); tokens outside H x W are the zero padding (index H*W = sentinel).
    Returns src (nW, N) int64 and rid (nW, N) int64.
    """
    Hp, Wp = -(-H // ws) * ws, -(-W // ws) * ws
    i, j = torch.arange(Hp), torch.arange(Wp)
    ho, wo = (i + shift) % Hp, (j + shift) % Wp
    src = ho[:, None] * W + wo[None, :]
    src = torch.where((ho[:, None] < H) & (wo[None, :] < W), src, torch.full_like(src, H * W))
    rh = (i >= Hp - ws).long() + (i >= Hp - shift).long() if shift > 0 else torch.zeros_like(i)
    rw = (j >= Wp - ws).long() + (j >= Wp - shift).long() if shift > 0 else torch.zeros_like(j)
    rid = rh[:, None] * 3 + rw[None, :]

    def part(t):
        return t.view(Hp // ws, ws, Wp // ws, ws).permute(0, 2, 1, 3).reshape(-1, ws * ws)

    return part(src), part(rid)


def relative_position_index(ws: int) -> Tensor:
    """backbone/swin.py:110-121 — (N, N) index into the (2ws-1)^2 bias table."""
    ys, xs = torch.meshgrid(torch.arange(ws), torch.arange(ws), indexing="ij")
    ys, xs = ys.reshape(-1), xs.reshape(-1)
    dy = ys[:, None] - ys[None, :] + ws - 1
    dx = xs[:, None] - xs[None, :] + ws - 1
    return dy * (2 * ws - 1) + dx


def window_attention(xn: Tensor, sd: SD, p: str, H: int, W: int, ws: int, shift: int,
                     nH: int, qk_scale: Optional[float] = None) -> Tensor:
    """norm1 output (B, H*W, C) -> attention branch output (B, H*W, C), incl. proj.

    backbone/swin.py:131-171 (WindowAttention.forward) inside :248-289.
    Padding slots are all-zero *inputs* to qkv, so their q/k/v equal the qkv bias and
    they take part as keys un-masked (swin.py:250-255: pad after norm1, no key mask).
    """
    B, L, C = xn.shape
    hd = C // nH
    scale = qk_scale or hd ** -0.5
    src, rid = window_layout(H, W, ws, shift)
    nW, N = src.shape
    xz = torch.cat([xn, xn.new_zeros(B, 1, C)], 1)                   # sentinel row = zero padding
    xw = xz[:, src.reshape(-1)].view(B * nW, N, C)
    qkv = _lin(xw, sd, p + ".qkv").view(B * nW, N, 3, nH, hd).permute(2, 0, 3, 1, 4)
    q, k, v = qkv[0] * scale, qkv[1], qkv[2]
    attn = q @ k.transpose(-1, -2)                                    # (B*nW, nH, N, N)
    table = sd[p + ".relative_position_bias_table"]                   # ((2ws-1)^2, nH)
    bias = table[relative_position_index(ws).reshape(-1)].view(N, N, nH).permute(2, 0, 1)
    attn = attn + bias[None]
    if shift > 0:
        m = (rid[:, :, None] != rid[:, None, :]).to(attn.dtype) * -100.0   # (nW, N, N); swin.py:438
        attn = (attn.view(B, nW, nH, N, N) + m[None, :, None]).view(B * nW, nH, N, N)
    attn = attn.softmax(-1)
    out = (attn @ v).transpose(1, 2).reshape(B * nW, N, C)
    out = _lin(out, sd, p + ".proj").view(B, nW * N, C)
    y = out.new_zeros(B, L + 1, C)
    y[:, src.reshape(-1)] = out                                       # padding slots land on the sentinel
    return y[:, :L]


def swin_block(x: Tensor, sd: SD, p: str, H: int, W: int, ws: int, shift: int, nH: int,
               qk_scale=None, branch_scale=None) -> Tensor:
    """backbone/swin.py:235-295.  branch_scale = None: eval mode, DropPath = identity; else two (B,) tensors, the per-sample
    multipliers timm's DropPath applies to the attention and to the MLP branch in training (0 or 1 / keep_prob: swin.py:279, 289)."""
    a = window_attention(_ln(x, sd, p + ".norm1"), sd, p + ".attn", H, W, ws, shift, nH, qk_scale)
    x = x + (a if branch_scale is None else a * branch_scale[0].view(-1, 1, 1))
    h = _lin(F.gelu(_lin(_ln(x, sd, p + ".norm2"), sd, p + ".mlp.fc1")), sd, p + ".mlp.fc2")
    return x + (h if branch_scale is None else h * branch_scale[1].view(-1, 1, 1))


def patch_merging(x: Tensor, sd: SD, p: str, H: int, W: int) -> Tensor:
    """backbone/swin.py:311-337: 2x2 gather in order (0,0),(1,0),(0,1),(1,1) -> LN(4C) -> 4C->2C."""
    B, L, C = x.shape
    x = x.view(B, H, W, C)
    x = F.pad(x, (0, 0, 0, W % 2, 0, H % 2))
    parts = [x[:, dy::2, dx::2] for dx in (0, 1) for dy in (0, 1)]
    x = torch.cat(parts, -1).reshape(B, -1, 4 * C)
    return F.linear(_ln(x, sd, p + ".norm"), sd[p + ".reduction.weight"])


def patch_embed(img: Tensor, sd: SD, p: str, patch: int = 4) -> Tuple[Tensor, int, int]:
    """backbone/swin.py:479-495: pad to x4, 4x4/4 conv, LayerNorm over channels -> (B, L, C)."""
    _, _, H, W = img.shape
    img = F.pad(img, (0, (-W) % patch, 0, (-H) % patch))
    x = F.conv2d(img, sd[p + ".proj.weight"], sd[p + ".proj.bias"], stride=patch)
    Hp, Wp = x.shape[-2:]
    x = x.flatten(2).transpose(1, 2)
    if p + ".norm.weight" in sd:
        x = _ln(x, sd, p + ".norm")
    return x, Hp, Wp


def swin_backbone(img: Tensor, sd: SD, cfg: SwinCfg, prefix: str = "backbone.") -> Dict[str, Tensor]:
    """backbone/swin.py:651-678 + D2 wrapper :743-758.  img (B,3,H,W) -> {"res2".."res5"} NCHW."""
    x, H, W = patch_embed(img, sd, prefix + "patch_embed", cfg.patch_size)
    outs = {}
    ws = cfg.window_size
    for s, depth in enumerate(cfg.depths):
        C = cfg.embed_dim * 2 ** s
        for i in range(depth):
            x = swin_block(x, sd, f"{prefix}layers.{s}.blocks.{i}", H, W, ws,
                           0 if i % 2 == 0 else ws // 2, cfg.num_heads[s], cfg.qk_scale)
        name = f"res{s + 2}"
        if name in cfg.out_features:
            o = _ln(x, sd, f"{prefix}norm{s}")
            outs[name] = o.view(-1, H, W, C).permute(0, 3, 1, 2).contiguous()
        if s < len(cfg.depths) - 1:
            x = patch_merging(x, sd, f"{prefix}layers.{s}.downsample", H, W)
            H, W = (H + 1) // 2, (W + 1) // 2
    return outs


# ----------------------------------------------------------------------------
# A5/A6: pixel decoder with multi-scale deformable attention
# ----------------------------------------------------------------------------
def position_embedding_sine(B: int, H: int, W: int, num_pos_feats: int = 128,
                            temperature: float = 10000.0) -> Tensor:
    """transformer_decoder/position_encoding.py:32-55 with mask=None, normalize=True -> (B, 2F, H, W)."""
    eps, scale = 1e-6, 2 * math.pi
    y = torch.arange(1, H + 1, dtype=torch.float32)
    x = torch.arange(1, W + 1, dtype=torch.float32)
    y = y / (float(H) + eps) * scale
    x = x / (float(W) + eps) * scale
    d = torch.arange(num_pos_feats, dtype=torch.float32)
    d = temperature ** (2 * torch.div(d, 2, rounding_mode="floor") / num_pos_feats)
    px, py = x[:, None] / d, y[:, None] / d                       # (W, F), (H, F)

    def interleave(t):
        return torch.stack((t[:, 0::2].sin(), t[:, 1::2].cos()), 2).flatten(1)

    px, py = interleave(px), interleave(py)
    pos = torch.cat((py[:, None, :].expand(H, W, -1), px[None, :, :].expand(H, W, -1)), 2)
    return pos.permute(2, 0, 1)[None].expand(B, -1, -1, -1)


def ms_deform_attn_core(value: Tensor, shapes: Sequence[Tuple[int, int]], loc: Tensor, w: Tensor) -> Tensor:
    """The arithmetic of the reference's only device kernel.

    pixel_decoder/ops/src/cuda/ms_deform_im2col_cuda.cuh:242-304 (forward kernel) and
    :38-89 (bilinear tap with per-tap bounds checks), restated with gathers:
      out[b,q,m,:] = sum_{l,p} w[b,q,m,l,p] * bilinear(value_l[b,:,m,:], (loc*(W_l,H_l) - 0.5))
    value (B,S,M,D); loc (B,Lq,M,L,P,2) as (x,y) in [0,1]; w (B,Lq,M,L,P) -> (B,Lq,M*D).
    """
    B, S, M, D = value.shape
    _, Lq, _, L, P, _ = loc.shape
    out = value.new_zeros(B, Lq, M, D)
    start = 0
    bidx = torch.arange(B)[:, None, None, None]
    midx = torch.arange(M)[None, None, :, None]
    for l, (Hl, Wl) in enumerate(shapes):
        v = value[:, start:start + Hl * Wl]                        # (B, Hl*Wl, M, D)
        start += Hl * Wl
        wim = loc[:, :, :, l, :, 0] * Wl - 0.5                      # (B,Lq,M,P)
        him = loc[:, :, :, l, :, 1] * Hl - 0.5
        inside = (him > -1) & (wim > -1) & (him < Hl) & (wim < Wl)
        h0, w0 = torch.floor(him), torch.floor(wim)
        lh, lw = him - h0, wim - w0
        h0, w0 = h0.long(), w0.long()
        acc = 0
        for dh, dw, cw in ((0, 0, (1 - lh) * (1 - lw)), (0, 1, (1 - lh) * lw),
                           (1, 0, lh * (1 - lw)), (1, 1, lh * lw)):
            hh, ww = h0 + dh, w0 + dw
            ok = inside & (hh >= 0) & (hh <= Hl - 1) & (ww >= 0) & (ww <= Wl - 1)
            idx = (hh.clamp(0, Hl - 1) * Wl + ww.clamp(0, Wl - 1))
            tap = v[bidx, idx, midx]                                # (B,Lq,M,P,D)
            acc = acc + tap * (cw * ok.to(v.dtype))[..., None]
        out = out + (acc * w[:, :, :, l, :, None]).sum(3)
    return out.reshape(B, Lq, M * D)


def ms_deform_attn(query: Tensor, ref: Tensor, src: Tensor, shapes, sd: SD, p: str,
                   M: int = 8, P: int = 4) -> Tensor:
    """pixel_decoder/ops/modules/ms_deform_attn.py:85-126 (reference_points last dim 2, no padding mask)."""
    B, Lq, C = query.shape
    L = len(shapes)
    value = _lin(src, sd, p + ".value_proj").view(B, -1, M, C // M)
    off = _lin(query, sd, p + ".sampling_offsets").view(B, Lq, M, L, P, 2)
    aw = _lin(query, sd, p + ".attention_weights").view(B, Lq, M, L * P).softmax(-1).view(B, Lq, M, L, P)
    norm = torch.tensor([[float(w_), float(h_)] for (h_, w_) in shapes])       # (L, 2) as (W, H)
    loc = ref[:, :, None, :, None, :] + off / norm[None, None, None, :, None, :]
    return _lin(ms_deform_attn_core(value, shapes, loc, aw), sd, p + ".output_proj")


def encoder_reference_points(shapes) -> Tensor:
    """pixel_decoder/msdeformattn.py:152-166 with all-valid masks (valid_ratios == 1): (1, sum HW, L, 2)."""
    pts = []
    for (H, W) in shapes:
        ys = (torch.arange(H, dtype=torch.float32) + 0.5) / H
        xs = (torch.arange(W, dtype=torch.float32) + 0.5) / W
        yy, xx = torch.meshgrid(ys, xs, indexing="ij")
        pts.append(torch.stack((xx.reshape(-1), yy.reshape(-1)), -1))
    pts = torch.cat(pts, 0)
    return pts[None, :, None, :].expand(1, -1, len(shapes), -1)


def pixel_decoder(feats: Dict[str, Tensor], sd: SD, cfg: HeadCfg,
                  prefix: str = "sem_seg_head.pixel_decoder.") -> Tuple[Tensor, Tensor, List[Tensor]]:
    """pixel_decoder/msdeformattn.py:336-386 (forward_features), eval mode.

    Returns mask_features (B,256,H/4,W/4), out[0], multi_scale_features [1/32, 1/16, 1/8].
    """
    names = list(cfg.transformer_in_features)[::-1]                # res5, res4, res3
    srcs, poss, shapes = [], [], []
    for i, n in enumerate(names):
        x = feats[n].float()
        s = F.conv2d(x, sd[f"{prefix}input_proj.{i}.0.weight"], sd[f"{prefix}input_proj.{i}.0.bias"])
        s = _gn(s, sd, f"{prefix}input_proj.{i}.1")
        B, C, H, W = s.shape
        shapes.append((H, W))
        srcs.append(s.flatten(2).transpose(1, 2))
        pe = position_embedding_sine(B, H, W, cfg.conv_dim // 2).flatten(2).transpose(1, 2)
        poss.append(pe + sd[prefix + "transformer.level_embed"][i].view(1, 1, -1))
    src, pos = torch.cat(srcs, 1), torch.cat(poss, 1)
    ref = encoder_reference_points(shapes)
    x = src
    for l in range(cfg.enc_layers):                                 # msdeformattn.py:132-142
        lp = f"{prefix}transformer.encoder.layers.{l}"
        x = _ln(x + ms_deform_attn(x + pos, ref, x, shapes, sd, lp + ".self_attn", cfg.nheads, cfg.n_points),
                sd, lp + ".norm1")
        x = _ln(x + _lin(F.relu(_lin(x, sd, lp + ".linear1")), sd, lp + ".linear2"), sd, lp + ".norm2")
    out, start = [], 0
    B = x.shape[0]
    for (H, W) in shapes:
        out.append(x[:, start:start + H * W].transpose(1, 2).reshape(B, -1, H, W))
        start += H * W
    # FPN with res2 (msdeformattn.py:369-379); GN after both convs, ReLU after the 3x3
    lat = _gn(F.conv2d(feats["res2"].float(), sd[prefix + "adapter_1.weight"]), sd, prefix + "adapter_1.norm")
    y = lat + F.interpolate(out[-1], size=lat.shape[-2:], mode="bilinear", align_corners=False)
    y = F.relu(_gn(F.conv2d(y, sd[prefix + "layer_1.weight"], padding=1), sd, prefix + "layer_1.norm"))
    out.append(y)
    mf = F.conv2d(out[-1], sd[prefix + "mask_features.weight"], sd[prefix + "mask_features.bias"])
    return mf, out[0], out[:3]


# ----------------------------------------------------------------------------
# A7/A8: task-conditioned masked-attention transformer decoder
# ----------------------------------------------------------------------------
def mha(q_in: Tensor, k_in: Tensor, v_in: Tensor, sd: SD, p: str, nheads: int,
        mask: Optional[Tensor] = None) -> Tensor:
    """torch.nn.MultiheadAttention forward (batch-first restatement), bool mask True = blocked.

    q_in (B,Lq,E), k_in/v_in (B,S,E), mask (B,Lq,S) bool shared by all heads
    (the reference repeats it over heads, oneformer_transformer_decoder.py:510).
    """
    E = q_in.shape[-1]
    Wi, bi = sd[p + ".in_proj_weight"], sd[p + ".in_proj_bias"]
    q = F.linear(q_in, Wi[:E], bi[:E])
    k = F.linear(k_in, Wi[E:2 * E], bi[E:2 * E])
    v = F.linear(v_in, Wi[2 * E:], bi[2 * E:])
    B, Lq, _ = q.shape
    S = k.shape[1]
    hd = E // nheads
    q = q.view(B, Lq, nheads, hd).transpose(1, 2) * hd ** -0.5
    k = k.view(B, S, nheads, hd).transpose(1, 2)
    v = v.view(B, S, nheads, hd).transpose(1, 2)
    a = q @ k.transpose(-1, -2)
    if mask is not None:
        a = a.masked_fill(mask[:, None], float("-inf"))
    o = (a.softmax(-1) @ v).transpose(1, 2).reshape(B, Lq, E)
    return _lin(o, sd, p + ".out_proj")


def _mlp(x: Tensor, sd: SD, p: str, n: int) -> Tensor:
    for i in range(n):
        x = _lin(x, sd, f"{p}.layers.{i}")
        if i < n - 1:
            x = F.relu(x)
    return x


def prediction_heads(out: Tensor, mf: Tensor, size, sd: SD, prefix: str):
    """transformer_decoder/oneformer_transformer_decoder.py:495-513.  out (B,Q,E), mf (B,C,H,W)."""
    d = _ln(out, sd, prefix + "decoder_norm")
    cls = _lin(d, sd, prefix + "class_embed")
    me = _mlp(d, sd, prefix + "mask_embed", 3)
    masks = torch.einsum("bqc,bchw->bqhw", me, mf)
    am = F.interpolate(masks, size=size, mode="bilinear", align_corners=False)
    am = (am.sigmoid().flatten(2) < 0.5)                           # (B,Q,S) True = blocked
    return cls, masks, am


def transformer_decoder(ms_feats: List[Tensor], mf: Tensor, tasks: Tensor, sd: SD, cfg: HeadCfg,
                        prefix: str = "sem_seg_head.predictor.", forced_masks=None, is_train: bool = False) -> Dict[str, object]:
    """transformer_decoder/oneformer_transformer_decoder.py:405-493 (dropout = identity).  is_train: the constructor flag
    of :273 / :477-480 -- `contrastive_logits` is then the PRE-loop query tensor (class-transformer output + task token)."""
    nh, E = cfg.nheads, cfg.hidden_dim
    B = mf.shape[0]
    src, pos, sizes = [], [], []
    for i, f in enumerate(ms_feats):
        H, W = f.shape[-2:]
        sizes.append((H, W))
        pos.append(position_embedding_sine(B, H, W, E // 2).flatten(2).transpose(1, 2))
        src.append(f.flatten(2).transpose(1, 2) + sd[prefix + "level_embed.weight"][i].view(1, 1, -1))
    qe = sd[prefix + "query_embed.weight"]                          # (Q, E)
    t = tasks[:, None, :]
    if cfg.use_task_norm:
        t = _ln(t, sd, prefix + "decoder_norm")
    # class_transformer: transformer.py:64-82 (0 encoder layers) + forward_post :268-297
    # NB the call at :434-437 passes the sine embedding as `src` (-> memory, i.e. the VALUES) and the
    # projected mask features as `pos_embed`: keys = PE + proj(mf), values = PE.
    H4, W4 = mf.shape[-2:]
    mpos = F.conv2d(mf, sd[prefix + "class_input_proj.weight"], sd[prefix + "class_input_proj.bias"])
    mpos = mpos.flatten(2).transpose(1, 2)
    mem = position_embedding_sine(B, H4, W4, E // 2).flatten(2).transpose(1, 2)
    qpos = qe[:-1][None].expand(B, -1, -1)
    tgt = t.expand(-1, qe.shape[0] - 1, -1) if cfg.use_task_norm else torch.zeros_like(qpos)
    for l in range(cfg.class_dec_layers):
        lp = f"{prefix}class_transformer.decoder.layers.{l}"
        qk = tgt + qpos
        tgt = _ln(tgt + mha(qk, qk, tgt, sd, lp + ".self_attn", nh), sd, lp + ".norm1")
        tgt = _ln(tgt + mha(tgt + qpos, mem + mpos, mem, sd, lp + ".multihead_attn", nh), sd, lp + ".norm2")
        tgt = _ln(tgt + _lin(F.relu(_lin(tgt, sd, lp + ".linear1")), sd, lp + ".linear2"), sd, lp + ".norm3")
    tgt = _ln(tgt, sd, prefix + "class_transformer.decoder.norm")
    out = torch.cat([tgt, t], 1)                                    # (B, Q, E): 149 queries + task token
    query_class = out                                               # :440 `out` (batch-first here; the reference permutes at :478)
    qpos = qe[None].expand(B, -1, -1)
    pc, pm, ams = [], [], []
    cls, masks, am = prediction_heads(out, mf, sizes[0], sd, prefix)
    pc.append(cls); pm.append(masks)
    for i in range(cfg.dec_layers):
        lvl = i % 3
        if forced_masks is not None:
            am = forced_masks[i]
        am = am & ~am.all(-1, keepdim=True)                          # :454 un-mask fully blocked rows
        ams.append(am)
        lp = f"{prefix}transformer_cross_attention_layers.{i}"
        out = _ln(out + mha(out + qpos, src[lvl] + pos[lvl], src[lvl], sd, lp + ".multihead_attn", nh, am),
                  sd, lp + ".norm")
        lp = f"{prefix}transformer_self_attention_layers.{i}"
        qk = out + qpos
        out = _ln(out + mha(qk, qk, out, sd, lp + ".self_attn", nh), sd, lp + ".norm")
        lp = f"{prefix}transformer_ffn_layers.{i}"
        out = _ln(out + _lin(F.relu(_lin(out, sd, lp + ".linear1")), sd, lp + ".linear2"), sd, lp + ".norm")
        cls, masks, am = prediction_heads(out, mf, sizes[(i + 1) % 3], sd, prefix)
        pc.append(cls); pm.append(masks)
    return {"pred_logits": pc[-1], "pred_masks": pm[-1],
            "aux_outputs": [{"pred_logits": a, "pred_masks": b} for a, b in zip(pc[:-1], pm[:-1])],
            "attn_masks": ams, "contrastive_logits": query_class if is_train else None}


# ----------------------------------------------------------------------------
# A10: meta-architecture glue, segmentation branch
# ----------------------------------------------------------------------------
def preprocess(images: List[Tensor], cfg: ModelCfg) -> Tensor:
    """oneformer_model.py:245-247: (x - mean) / std, then zero-pad bottom/right to a multiple of 32."""
    mean = torch.tensor(cfg.pixel_mean).view(3, 1, 1)
    std = torch.tensor(cfg.pixel_std).view(3, 1, 1)
    imgs = [(im.float() - mean) / std for im in images]
    d = cfg.size_divisibility
    H = max(-(-im.shape[1] // d) * d for im in imgs)
    W = max(-(-im.shape[2] // d) * d for im in imgs)
    return torch.stack([F.pad(im, (0, W - im.shape[2], 0, H - im.shape[1])) for im in imgs])


def task_embedding(tasks: List[str], sd: SD, cfg: ModelCfg) -> Tensor:
    """oneformer_model.py:249-251: token ids .float() -> MLP(77, 256, 256, 2)."""
    tok = torch.stack([task_tokens(t, cfg.task_seq_len) for t in tasks]).float()
    return _mlp(tok, sd, "task_mlp", 2)


def oneformer_forward(batched_inputs: List[dict], sd: SD, cfg: ModelCfg, upsample: bool = True, forced_masks=None):
    """oneformer_model.py:244-263 (segmentation branch up to the mask upsample; no post-processing)."""
    x = preprocess([b["left_image"] for b in batched_inputs], cfg)
    tasks = task_embedding([b["task"] for b in batched_inputs], sd, cfg)
    feats = swin_backbone(x, sd, cfg.swin)
    mf, _, ms = pixel_decoder(feats, sd, cfg.head)
    out = transformer_decoder(ms, mf, tasks, sd, cfg.head, forced_masks=forced_masks)
    if upsample:
        out["pred_masks_up"] = F.interpolate(out["pred_masks"], size=x.shape[-2:], mode="bilinear",
                                             align_corners=False)
    return out


def synthetic_loss(out: dict) -> Tensor:
    """The bench/test training objective (the reference ships no criterion):
    mean-square of logits and masks over the final and (x0.1) the nine auxiliary predictions."""
    loss = out["pred_logits"].float().square().mean() + out["pred_masks"].float().square().mean()
    for a in out["aux_outputs"]:
        loss = loss + 0.1 * (a["pred_logits"].float().square().mean() + a["pred_masks"].float().square().mean())
    return loss


# ----------------------------------------------------------------------------
# parameter shapes (so oracle, goldens and product agree without building modules)
# ----------------------------------------------------------------------------
def swin_param_shapes(cfg: SwinCfg, prefix: str = "backbone.") -> Dict[str, Tuple[int, ...]]:
    s: Dict[str, Tuple[int, ...]] = {}
    C0, ws = cfg.embed_dim, cfg.window_size
    s[prefix + "patch_embed.proj.weight"] = (C0, 3, cfg.patch_size, cfg.patch_size)
    s[prefix + "patch_embed.proj.bias"] = (C0,)
    s[prefix + "patch_embed.norm.weight"] = (C0,)
    s[prefix + "patch_embed.norm.bias"] = (C0,)
    for st, depth in enumerate(cfg.depths):
        C, nH = C0 * 2 ** st, cfg.num_heads[st]
        hid = int(C * cfg.mlp_ratio)
        for i in range(depth):
            p = f"{prefix}layers.{st}.blocks.{i}."
            for n in ("norm1", "norm2"):
                s[p + n + ".weight"] = (C,); s[p + n + ".bias"] = (C,)
            s[p + "attn.relative_position_bias_table"] = ((2 * ws - 1) ** 2, nH)
            s[p + "attn.qkv.weight"] = (3 * C, C); s[p + "attn.qkv.bias"] = (3 * C,)
            s[p + "attn.proj.weight"] = (C, C); s[p + "attn.proj.bias"] = (C,)
            s[p + "mlp.fc1.weight"] = (hid, C); s[p + "mlp.fc1.bias"] = (hid,)
            s[p + "mlp.fc2.weight"] = (C, hid); s[p + "mlp.fc2.bias"] = (C,)
        if st < len(cfg.depths) - 1:
            p = f"{prefix}layers.{st}.downsample."
            s[p + "reduction.weight"] = (2 * C, 4 * C)
            s[p + "norm.weight"] = (4 * C,); s[p + "norm.bias"] = (4 * C,)
        s[f"{prefix}norm{st}.weight"] = (C,); s[f"{prefix}norm{st}.bias"] = (C,)
    return s


def head_param_shapes(cfg: HeadCfg, in_channels: Dict[str, int]) -> Dict[str, Tuple[int, ...]]:
    s: Dict[str, Tuple[int, ...]] = {}
    D, E = cfg.conv_dim, cfg.hidden_dim
    p = "sem_seg_head.pixel_decoder."
    for i, n in enumerate(list(cfg.transformer_in_features)[::-1]):
        s[f"{p}input_proj.{i}.0.weight"] = (D, in_channels[n], 1, 1); s[f"{p}input_proj.{i}.0.bias"] = (D,)
        s[f"{p}input_proj.{i}.1.weight"] = (D,); s[f"{p}input_proj.{i}.1.bias"] = (D,)
    L = len(cfg.transformer_in_features)
    s[p + "transformer.level_embed"] = (L, D)
    for l in range(cfg.enc_layers):
        q = f"{p}transformer.encoder.layers.{l}."
        s[q + "self_attn.sampling_offsets.weight"] = (cfg.nheads * L * cfg.n_points * 2, D)
        s[q + "self_attn.sampling_offsets.bias"] = (cfg.nheads * L * cfg.n_points * 2,)
        s[q + "self_attn.attention_weights.weight"] = (cfg.nheads * L * cfg.n_points, D)
        s[q + "self_attn.attention_weights.bias"] = (cfg.nheads * L * cfg.n_points,)
        for n in ("value_proj", "output_proj"):
            s[q + f"self_attn.{n}.weight"] = (D, D); s[q + f"self_attn.{n}.bias"] = (D,)
        s[q + "linear1.weight"] = (cfg.enc_ffn, D); s[q + "linear1.bias"] = (cfg.enc_ffn,)
        s[q + "linear2.weight"] = (D, cfg.enc_ffn); s[q + "linear2.bias"] = (D,)
        for n in ("norm1", "norm2"):
            s[q + n + ".weight"] = (D,); s[q + n + ".bias"] = (D,)
    s[p + "mask_features.weight"] = (cfg.mask_dim, D, 1, 1); s[p + "mask_features.bias"] = (cfg.mask_dim,)
    s[p + "adapter_1.weight"] = (D, in_channels["res2"], 1, 1)
    s[p + "adapter_1.norm.weight"] = (D,); s[p + "adapter_1.norm.bias"] = (D,)
    s[p + "layer_1.weight"] = (D, D, 3, 3)
    s[p + "layer_1.norm.weight"] = (D,); s[p + "layer_1.norm.bias"] = (D,)
    p = "sem_seg_head.predictor."

    def mha_shapes(q):
        s[q + ".in_proj_weight"] = (3 * E, E); s[q + ".in_proj_bias"] = (3 * E,)
        s[q + ".out_proj.weight"] = (E, E); s[q + ".out_proj.bias"] = (E,)

    def ln(q):
        s[q + ".weight"] = (E,); s[q + ".bias"] = (E,)

    def ffn(q):
        s[q + ".linear1.weight"] = (cfg.dim_feedforward, E); s[q + ".linear1.bias"] = (cfg.dim_feedforward,)
        s[q + ".linear2.weight"] = (E, cfg.dim_feedforward); s[q + ".linear2.bias"] = (E,)

    for l in range(cfg.class_dec_layers):
        q = f"{p}class_transformer.decoder.layers.{l}"
        mha_shapes(q + ".self_attn"); mha_shapes(q + ".multihead_attn"); ffn(q)
        for n in ("norm1", "norm2", "norm3"):
            ln(f"{q}.{n}")
    ln(p + "class_transformer.decoder.norm")
    for i in range(cfg.dec_layers):
        mha_shapes(f"{p}transformer_self_attention_layers.{i}.self_attn")
        ln(f"{p}transformer_self_attention_layers.{i}.norm")
        mha_shapes(f"{p}transformer_cross_attention_layers.{i}.multihead_attn")
        ln(f"{p}transformer_cross_attention_layers.{i}.norm")
        ffn(f"{p}transformer_ffn_layers.{i}")
        ln(f"{p}transformer_ffn_layers.{i}.norm")
    ln(p + "decoder_norm")
    s[p + "query_embed.weight"] = (cfg.num_queries, E)
    s[p + "level_embed.weight"] = (3, E)
    s[p + "class_input_proj.weight"] = (E, cfg.mask_dim, 1, 1); s[p + "class_input_proj.bias"] = (E,)
    s[p + "class_embed.weight"] = (cfg.num_classes + 1, E); s[p + "class_embed.bias"] = (cfg.num_classes + 1,)
    for i in range(3):
        s[f"{p}mask_embed.layers.{i}.weight"] = (cfg.mask_dim if i == 2 else E, E)
        s[f"{p}mask_embed.layers.{i}.bias"] = (cfg.mask_dim if i == 2 else E,)
    return s


def model_param_shapes(cfg: ModelCfg) -> Dict[str, Tuple[int, ...]]:
    s = swin_param_shapes(cfg.swin)
    ch = {f"res{i + 2}": cfg.swin.embed_dim * 2 ** i for i in range(4)}
    s.update(head_param_shapes(cfg.head, ch))
    E = cfg.head.hidden_dim
    s["task_mlp.layers.0.weight"] = (E, cfg.task_seq_len); s["task_mlp.layers.0.bias"] = (E,)
    s["task_mlp.layers.1.weight"] = (E, E); s["task_mlp.layers.1.bias"] = (E,)
    return s
