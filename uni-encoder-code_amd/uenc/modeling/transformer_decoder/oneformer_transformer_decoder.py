"""Task-conditioned masked-attention transformer decoder on HIP kernels.

Counterpart of reference model/modeling/transformer_decoder/oneformer_transformer_decoder.py
(+ transformer.py for the class transformer): same class / registry names, constructor arguments and
parameter names (`class_transformer.decoder.layers.{i}.{self_attn,multihead_attn}.in_proj_weight`,
`transformer_{self,cross}_attention_layers.{i}`, `transformer_ffn_layers.{i}`, `decoder_norm`,
`query_embed`, `level_embed`, `class_input_proj`, `class_embed`, `mask_embed.layers.{i}`), same
`forward(x, mask_features, tasks, mask=None) -> {"pred_logits", "pred_masks", "aux_outputs",
"contrastive_logits"}`.

Differences in how it runs: batch-first `(B, L, E)` tokens instead of `(L, B, E)`; every projection /
FFN is a bf16 MFMA GEMM with fused bias / ReLU / residual epilogues; post-norm `LN(x + f(x))` is one
LayerNorm kernel reading the GEMM's residual-fused output; the mask einsum is a batched NT GEMM over
channels-last mask features; attention cores go through `ops.attention` (HIP flash-style kernels).
Training mode: the class transformer's dropout (`ONE_FORMER.DROPOUT`, 0.1) is applied as in the reference -- on the attention
probabilities inside the HIP attention kernels (hash-derived keep-mask) and as dropout1 / 2 / 3 + the FFN's inner dropout
(transformer.py:249-260, 283-296); the nine masked-attention layers have rate 0.0 hard-coded in the reference (:326-344).
Post-norm only (`PRE_NORM: False` in every shipped config).
"""
import logging
from typing import Optional

import torch
from torch import nn
from torch.nn import functional as F

from ... import kernels as K
from ... import ops
from ...d2 import Conv2d, Registry, configurable
from .position_encoding import PositionEmbeddingSine

TRANSFORMER_DECODER_REGISTRY = Registry("TRANSFORMER_MODULE")
TRANSFORMER_DECODER_REGISTRY.__doc__ = "Registry for transformer module in OneFormer."


def build_transformer_decoder(cfg, in_channels, mask_classification=True):
    name = cfg.MODEL.ONE_FORMER.TRANSFORMER_DECODER_NAME
    return TRANSFORMER_DECODER_REGISTRY.get(name)(cfg, in_channels, mask_classification)


class MultiheadAttention(nn.Module):
    """nn.MultiheadAttention's parameters (in_proj_weight (3E, E), in_proj_bias, out_proj) on HIP kernels.

    Batch-first: query (B, Lq, E), key / value (B, S, E); `attn_mask` (B, Lq, S) bool, True = blocked,
    shared by all heads (the reference repeats one mask over heads, oneformer_transformer_decoder.py:510).
    """

    def __init__(self, embed_dim, num_heads, dropout=0.0):
        super().__init__()
        self.embed_dim, self.num_heads = embed_dim, num_heads
        self.dropout = float(dropout)           # on the attention probabilities, training mode only (nn.MultiheadAttention semantics)
        self.last_seed = None                   # seed of the last training forward (tests rebuild the keep-mask from it)
        self.in_proj_weight = nn.Parameter(torch.empty(3 * embed_dim, embed_dim))
        self.in_proj_bias = nn.Parameter(torch.zeros(3 * embed_dim))
        self.out_proj = nn.Linear(embed_dim, embed_dim)
        nn.init.xavier_uniform_(self.in_proj_weight)

    def forward(self, query, key, value, attn_mask=None, residual=None):
        E = self.embed_dim
        W, b = self.in_proj_weight, self.in_proj_bias
        if query is key:
            qk = ops.linear(query, W, b, rows=(0, 2 * E))
            q, k = qk[..., :E], qk[..., E:]
        else:
            q = ops.linear(query, W, b, rows=(0, E))
            k = ops.linear(key, W, b, rows=(E, 2 * E))
        v = ops.linear(value, W, b, rows=(2 * E, 3 * E))
        if self.training and self.dropout > 0.0:
            self.last_seed = int(torch.randint(0, 2 ** 31 - 1, (1,)))      # torch's CPU generator: no device sync
            o = ops.attention(q, k, v, self.num_heads, attn_mask, self.dropout, self.last_seed)
        else:
            o = ops.attention(q, k, v, self.num_heads, attn_mask)
        if residual is not None:
            return ops.linear(o, self.out_proj.weight, self.out_proj.bias, residual=residual)
        return ops.linear(o, self.out_proj.weight, self.out_proj.bias, out_dtype=torch.float32)


def _ln(m: nn.LayerNorm, x, **kw):
    return ops.layer_norm(x, m.weight, m.bias, eps=m.eps, **kw)


class SelfAttentionLayer(nn.Module):
    def __init__(self, d_model, nhead, dropout=0.0, activation="relu", normalize_before=False):
        super().__init__()
        assert not normalize_before, "pre-norm is not used by any shipped config"
        self.self_attn = MultiheadAttention(d_model, nhead, dropout=dropout)
        self.norm = nn.LayerNorm(d_model)

    def forward(self, tgt, tgt_mask=None, tgt_key_padding_mask=None, query_pos=None):
        qk = tgt if query_pos is None else tgt + query_pos
        return _ln(self.norm, self.self_attn(qk, qk, tgt, attn_mask=tgt_mask, residual=tgt))


class CrossAttentionLayer(nn.Module):
    def __init__(self, d_model, nhead, dropout=0.0, activation="relu", normalize_before=False):
        super().__init__()
        assert not normalize_before
        self.multihead_attn = MultiheadAttention(d_model, nhead, dropout=dropout)
        self.norm = nn.LayerNorm(d_model)

    def forward(self, tgt, memory, memory_mask=None, memory_key_padding_mask=None, pos=None, query_pos=None, key_in=None):
        q = tgt if query_pos is None else tgt + query_pos
        k = key_in if key_in is not None else (memory if pos is None else memory + pos)
        return _ln(self.norm, self.multihead_attn(q, k, memory, attn_mask=memory_mask, residual=tgt))


class FFNLayer(nn.Module):
    def __init__(self, d_model, dim_feedforward=2048, dropout=0.0, activation="relu", normalize_before=False):
        super().__init__()
        assert not normalize_before and activation == "relu"
        self.linear1 = nn.Linear(d_model, dim_feedforward)
        self.linear2 = nn.Linear(dim_feedforward, d_model)
        self.norm = nn.LayerNorm(d_model)

    def forward(self, tgt):
        h = ops.mlp(tgt, [self.linear1.weight, self.linear1.bias, self.linear2.weight, self.linear2.bias], act="relu", residual=tgt)
        return _ln(self.norm, h)


class MLP(nn.Module):
    """Linear -> ReLU chains (mask_embed, task_mlp); same `layers.{i}` names as the reference (:211-223)."""

    def __init__(self, input_dim, hidden_dim, output_dim, num_layers):
        super().__init__()
        self.num_layers = num_layers
        h = [hidden_dim] * (num_layers - 1)
        self.layers = nn.ModuleList(nn.Linear(n, k) for n, k in zip([input_dim] + h, h + [output_dim]))

    def forward(self, x, out_dtype=torch.float32):
        params = []
        for l in self.layers:
            params += [l.weight, l.bias]
        return ops.mlp(x, params, act="relu", out_dtype=out_dtype)


class TransformerDecoderLayer(nn.Module):
    """Post-norm DETR decoder layer of the class transformer (reference transformer.py:237-297)."""

    def __init__(self, d_model, nhead, dim_feedforward=2048, dropout=0.1, activation="relu", normalize_before=False):
        super().__init__()
        assert not normalize_before and activation == "relu"
        self.self_attn = MultiheadAttention(d_model, nhead, dropout=dropout)
        self.multihead_attn = MultiheadAttention(d_model, nhead, dropout=dropout)
        self.linear1 = nn.Linear(d_model, dim_feedforward)
        self.linear2 = nn.Linear(dim_feedforward, d_model)
        self.norm1, self.norm2, self.norm3 = nn.LayerNorm(d_model), nn.LayerNorm(d_model), nn.LayerNorm(d_model)
        self.dropout_p = float(dropout)         # dropout, dropout1 / 2 / 3 of the reference (transformer.py:251, 258-260), training only
        self._masks = []                        # keep-masks of the last training forward, in order (tests)

    def _drop(self, t):
        """Inverted dropout with an explicit keep-mask (torch.nn.Dropout semantics: x * mask / keep_prob)."""
        keep = 1.0 - self.dropout_p
        mask = torch.rand(t.shape, device=t.device) < keep
        self._masks.append(mask)
        return t * (mask.to(t.dtype) / keep)

    def forward(self, tgt, memory, key_in, query_pos):
        if self.training and self.dropout_p > 0.0:
            # training: the residual / FFN fusions are split where the reference applies dropout1 / 2 / 3 and the FFN's inner dropout
            # (transformer.py:283-296); the attention-probability dropout runs inside the attention kernels
            self._masks = []
            qk = tgt + query_pos
            tgt = _ln(self.norm1, tgt + self._drop(self.self_attn(qk, qk, tgt)))
            tgt = _ln(self.norm2, tgt + self._drop(self.multihead_attn(tgt + query_pos, key_in, memory)))
            h = self._drop(F.relu(ops.linear(tgt, self.linear1.weight, self.linear1.bias, out_dtype=torch.float32)))
            h = ops.linear(h, self.linear2.weight, self.linear2.bias, out_dtype=torch.float32)
            return _ln(self.norm3, tgt + self._drop(h))
        qk = tgt + query_pos
        tgt = _ln(self.norm1, self.self_attn(qk, qk, tgt, residual=tgt))
        tgt = _ln(self.norm2, self.multihead_attn(tgt + query_pos, key_in, memory, residual=tgt))
        h = ops.mlp(tgt, [self.linear1.weight, self.linear1.bias, self.linear2.weight, self.linear2.bias], act="relu", residual=tgt)
        return _ln(self.norm3, h)


class TransformerDecoder(nn.Module):
    def __init__(self, layer_args, num_layers, norm=None):
        super().__init__()
        self.layers = nn.ModuleList([TransformerDecoderLayer(*layer_args) for _ in range(num_layers)])
        self.num_layers, self.norm = num_layers, norm


class TransformerEncoder(nn.Module):
    def __init__(self, num_layers, norm=None):
        super().__init__()
        if num_layers != 0:
            raise NotImplementedError("ENC_LAYERS is 0 in every shipped config (oneformer_R50_bs16_90k.yaml:33)")
        self.layers = nn.ModuleList()
        self.num_layers, self.norm = num_layers, norm


class Transformer(nn.Module):
    """The `class_transformer` (reference transformer.py:22-82): 0 encoder layers + N post-norm decoder layers."""

    def __init__(self, d_model=512, nhead=8, num_encoder_layers=6, num_decoder_layers=6, dim_feedforward=2048, dropout=0.1,
                 activation="relu", normalize_before=False, return_intermediate_dec=False):
        super().__init__()
        self.encoder = TransformerEncoder(num_encoder_layers, nn.LayerNorm(d_model) if normalize_before else None)
        self.decoder = TransformerDecoder((d_model, nhead, dim_feedforward, dropout, activation, normalize_before),
                                          num_decoder_layers, nn.LayerNorm(d_model))
        for p in self.parameters():
            if p.dim() > 1:
                nn.init.xavier_uniform_(p)
        self.d_model, self.nhead = d_model, nhead

    def forward(self, memory, key_in, query_pos, tgt):
        """memory (values) / key_in (B, S, E); query_pos, tgt (B, Q, E) -> (B, Q, E)."""
        for layer in self.decoder.layers:
            tgt = layer(tgt, memory, key_in, query_pos)
        return _ln(self.decoder.norm, tgt)


@TRANSFORMER_DECODER_REGISTRY.register()
class ContrastiveMultiScaleMaskedTransformerDecoder(nn.Module):
    _version = 2

    def _load_from_state_dict(self, state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys, error_msgs):
        version = local_metadata.get("version", None)
        if version is None or version < 2:
            for k in list(state_dict.keys()):
                if "static_query" in k:       # legacy key of older checkpoints (reference :231-252)
                    state_dict[k.replace("static_query", "query_feat")] = state_dict.pop(k)
                    logging.getLogger(__name__).warning("converted legacy key %s", k)
        super()._load_from_state_dict(state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys, error_msgs)

    @configurable
    def __init__(self, in_channels, mask_classification=True, *, num_classes: int, hidden_dim: int, num_queries: int,
                 nheads: int, dropout: float, dim_feedforward: int, enc_layers: int, is_train: bool, dec_layers: int,
                 class_dec_layers: int, pre_norm: bool, mask_dim: int, enforce_input_project: bool, use_task_norm: bool):
        super().__init__()
        assert mask_classification, "Only support mask classification model"
        self.mask_classification, self.is_train, self.use_task_norm = mask_classification, is_train, use_task_norm
        self.pe_layer = PositionEmbeddingSine(hidden_dim // 2, normalize=True)
        self.class_transformer = Transformer(d_model=hidden_dim, dropout=dropout, nhead=nheads, dim_feedforward=dim_feedforward,
                                             num_encoder_layers=enc_layers, num_decoder_layers=class_dec_layers,
                                             normalize_before=pre_norm, return_intermediate_dec=False)
        self.num_heads, self.num_layers = nheads, dec_layers
        self.transformer_self_attention_layers = nn.ModuleList()
        self.transformer_cross_attention_layers = nn.ModuleList()
        self.transformer_ffn_layers = nn.ModuleList()
        for _ in range(self.num_layers):
            self.transformer_self_attention_layers.append(SelfAttentionLayer(hidden_dim, nheads, 0.0, normalize_before=pre_norm))
            self.transformer_cross_attention_layers.append(CrossAttentionLayer(hidden_dim, nheads, 0.0, normalize_before=pre_norm))
            self.transformer_ffn_layers.append(FFNLayer(hidden_dim, dim_feedforward, 0.0, normalize_before=pre_norm))
        self.decoder_norm = nn.LayerNorm(hidden_dim)
        self.num_queries = num_queries
        self.query_embed = nn.Embedding(num_queries, hidden_dim)
        self.num_feature_levels = 3
        self.level_embed = nn.Embedding(self.num_feature_levels, hidden_dim)
        self.input_proj = nn.ModuleList()
        for _ in range(self.num_feature_levels):
            if in_channels != hidden_dim or enforce_input_project:
                self.input_proj.append(Conv2d(in_channels, hidden_dim, kernel_size=1))
            else:
                self.input_proj.append(nn.Sequential())
        self.class_input_proj = Conv2d(in_channels, hidden_dim, kernel_size=1)
        if self.mask_classification:
            self.class_embed = nn.Linear(hidden_dim, num_classes + 1)
        self.mask_embed = MLP(hidden_dim, hidden_dim, mask_dim, 3)
        # test hook: a list of (B, Q, S) bool masks replacing the thresholded predictions, so that parity tests can
        # pin the discrete attention-mask path to the reference's and measure the continuous arithmetic alone
        self._mem_cache = {}                 # (B, H/4, W/4, device) -> sine-embedding tokens of the 1/4 map (see forward)
        self.forced_attn_masks = None

    @classmethod
    def from_config(cls, cfg, in_channels, mask_classification):
        of = cfg.MODEL.ONE_FORMER
        assert of.DEC_LAYERS >= 1
        return {"in_channels": in_channels, "mask_classification": mask_classification,
                "num_classes": cfg.MODEL.SEM_SEG_HEAD.NUM_CLASSES, "hidden_dim": of.HIDDEN_DIM,
                "num_queries": of.NUM_OBJECT_QUERIES, "nheads": of.NHEADS, "dim_feedforward": of.DIM_FEEDFORWARD,
                "dec_layers": of.DEC_LAYERS - 1, "class_dec_layers": of.CLASS_DEC_LAYERS, "enc_layers": of.ENC_LAYERS,
                "dropout": of.DROPOUT, "pre_norm": of.PRE_NORM, "enforce_input_project": of.ENFORCE_INPUT_PROJ,
                "is_train": cfg.MODEL.IS_TRAIN, "mask_dim": cfg.MODEL.SEM_SEG_HEAD.MASK_DIM,
                "use_task_norm": of.USE_TASK_NORM}

    @staticmethod
    def _tok(x):
        B, C, H, W = x.shape
        return x.permute(0, 2, 3, 1).reshape(B, H * W, C)

    def forward(self, x, mask_features, tasks, mask=None):
        assert len(x) == self.num_feature_levels
        del mask                                             # unused in the reference as well (:413)
        B = mask_features.shape[0]
        dev = mask_features.device
        src, kin, size_list = [], [], []
        for i in range(self.num_feature_levels):
            H, W = x[i].shape[-2:]
            size_list.append((H, W))
            t = self._tok(x[i])
            ip = self.input_proj[i]
            if not (isinstance(ip, nn.Sequential) and len(ip) == 0):     # identity when in_channels == hidden_dim (:358-364)
                t = ops.linear(t, ip.weight, ip.bias, out_dtype=torch.float32)
            s = t + self.level_embed.weight[i].view(1, 1, -1)
            src.append(s)
            kin.append(s + self.pe_layer.tokens(B, H, W, dev))
        Q = self.num_queries
        qe = self.query_embed.weight[None].expand(B, -1, -1)
        t_tok = tasks[:, None, :]
        if self.use_task_norm:
            t_tok = _ln(self.decoder_norm, t_tok)

        # class transformer: values = sine embedding of the 1/4 map, keys = that + class_input_proj(mask_features)
        # (the reference passes PE as `src` and the projection as `pos_embed`, :434-437)
        H4, W4 = mask_features.shape[-2:]
        mf32 = self._tok(mask_features.float())
        if not mf32.is_contiguous():
            mf32 = mf32.contiguous()
        # one materialised map per geometry, kept across steps (it depends on the shape only): every layer's value projection reuses its
        # bf16 copy (ops._twin), and neither the 268 MB expansion nor its cast is redone every step
        mkey = (B, H4, W4, str(dev))
        mem = self._mem_cache.get(mkey)
        if mem is None:
            self._mem_cache.clear()
            mem = self.pe_layer.tokens(B, H4, W4, dev).contiguous()
            self._mem_cache[mkey] = mem
            if not K.EXACT:
                # its bf16 operand copy lives as long as the map itself (a twin registered for the per-call reshaped view died with that
                # view after every backward, and the 268 MB map was re-cast every step)
                ops._register_twin(mem, K.cast_bf16(mem.view(-1, mem.shape[-1])))
        # (bf16 result: the sum only feeds the two key projections of the class transformer)
        key_in = ops.linear(mf32, self.class_input_proj.weight, self.class_input_proj.bias, residual=mem, out_dtype=K.adt())
        tgt = t_tok.expand(-1, Q - 1, -1) if self.use_task_norm else torch.zeros_like(qe[:, :-1])
        out_t = self.class_transformer(mem, key_in, qe[:, :-1], tgt)
        output = torch.cat([out_t, t_tok], 1)
        query_class = output                                 # reference :440, 477-478: the PRE-loop queries (batch-first here)

        mf16 = None if K.EXACT else ops._twin(mf32)          # the bf16 copy the class_input_proj GEMM above made of the same map
        if mf16 is None:
            mf16 = mf32.detach().to(K.adt())
        # (B, C, HW) bf16 operand of the mask-embedding gradient GEMM: LDS-tiled cast + transpose per image (a strided ATen copy of the
        # 134 MB map took 0.28 ms)
        mf16_chw = torch.empty((B, mf32.shape[2], mf32.shape[1]), dtype=K.adt(), device=mf32.device)
        for b in range(B):
            K.cast_transpose_bf16(mf32[b].detach(), out=mf16_chw[b])
        mes = []                                             # mask embeddings of all heads: their einsums share one autograd node
        predictions_class, predictions_mask = [], []         # (predictions_class collects the heads' normalised queries: see below)
        cls, msk, attn_mask = self.forward_prediction_heads(output, (mf16, H4, W4, mes), size_list[0])
        predictions_class.append(cls); predictions_mask.append(msk)
        for i in range(self.num_layers):
            lvl = i % self.num_feature_levels
            if self.forced_attn_masks is not None:
                attn_mask = self.forced_attn_masks[i]
                attn_mask = attn_mask & ~attn_mask.all(-1, keepdim=True)     # un-block fully blocked rows (:454)
            output = self.transformer_cross_attention_layers[i](output, src[lvl], memory_mask=attn_mask, query_pos=qe,
                                                                key_in=kin[lvl])
            output = self.transformer_self_attention_layers[i](output, query_pos=qe)
            output = self.transformer_ffn_layers[i](output)
            cls, msk, attn_mask = self.forward_prediction_heads(output, (mf16, H4, W4, mes),
                                                                size_list[(i + 1) % self.num_feature_levels])
            predictions_class.append(cls); predictions_mask.append(msk)
        assert len(predictions_class) == self.num_layers + 1
        # class logits of ALL heads in one GEMM: nothing inside the loop consumes them (only the mask logits steer the next layer), so
        # the ten 300 x 20 x 256 launches (+ their padded-N dgrad / wgrad / bias glue in the backward) become one 3000-row launch
        nq = predictions_class[0].shape[1]
        allc = ops.linear(torch.cat(predictions_class, 1), self.class_embed.weight, self.class_embed.bias, out_dtype=torch.float32)
        predictions_class = [allc[:, i * nq:(i + 1) * nq] for i in range(len(predictions_class))]
        # the eager mask logits become differentiable here: one node for all heads (ops.MaskHeadsFn)
        predictions_mask = [m.view(B, -1, H4, W4) for m in ops.mask_heads(mf32, mf16_chw, [m.view(B, -1, H4 * W4) for m in predictions_mask], mes)]
        return {"contrastive_logits": query_class if self.is_train else None,
                "pred_logits": predictions_class[-1], "pred_masks": predictions_mask[-1],
                "aux_outputs": [{"pred_logits": a, "pred_masks": b} for a, b in zip(predictions_class[:-1], predictions_mask[:-1])]}

    def forward_prediction_heads(self, output, mf, attn_mask_target_size):
        mf16, H4, W4, mes = mf
        d = _ln(self.decoder_norm, output)
        outputs_class = d                                    # its class logits are computed with all other heads' at the end of forward
        me = self.mask_embed(d, out_dtype=torch.bfloat16).contiguous()
        B, Q, _ = me.shape
        mes.append(me)
        outputs_mask = ops.mask_logits_eager(me, mf16).view(B, Q, H4, W4)        # detached: made differentiable at the end of forward
        # (B, Q, S) True = blocked, shared by all heads: resize + sigmoid < 0.5 + the all-blocked-row fix of :454 in one kernel
        am = K.attn_mask(outputs_mask.detach(), attn_mask_target_size)
        return outputs_class, outputs_mask, am
