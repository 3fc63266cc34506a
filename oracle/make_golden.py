"""Generate tests/golden/*.npz from the reference's own modules (build container only).

TEST INFRASTRUCTURE.  Run:  python -m oracle.make_golden
Imports the reference's hot-path files from /root/reference on CPU through
`oracle/ref_loader.py`, fills them with the name-hashed weights of `oracle/fill.py`
and stores inputs + outputs (never weights, never source) as small fp32 fixtures.
The fixtures travel to the GPU box; the reference does not.

Fixture list:
  F1 swin_pair_ws7     W-MSA + SW-MSA block pair, C=96 nH=3 ws=7 on 24x40 (pads to 28x42)
  F2 swin_pair_ws12    same, C=192 nH=6 ws=12 on 20x30 (pads to 24x36)
  F3 swin_t_96x160     full Swin-T on 1x3x96x160 -> res2..res5
  F4 patch_merging     odd H, W
  F5 msdeform_core     value, loc, w -> out and the three gradients
  F6 pixel_decoder     Swin-shaped features @64x96 -> mask_features + 3 maps
  F7 transformer_decoder  F6 outputs + task vector -> logits, masks, aux, attention masks
  F8 pos_embed_sine    5x7
  F9 task_tokens       tokenizer ids of the three task prompts + task_mlp output
  F10 model_fwd_bwd    small full model: loss and selected parameter gradients
  F11 decoder_contrastive  F7's inputs through the decoder built with is_train=True: `contrastive_logits` and gradients of a loss on it
  F12 swin_ape         SwinTransformer with APE = True and non-zero drop rates (eval mode): features and the embedding's gradient

`python -m oracle.make_golden NAME [NAME ...]` regenerates only the named fixtures (F11: decoder_contrastive).
"""
import json
import os
import sys
import types
import warnings

import numpy as np
import torch

from . import fill, ref_loader
from . import torch_ref as T

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


def _np(t):
    return t.detach().cpu().numpy()


def _save(name, **arrs):
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **{k: (_np(v) if isinstance(v, torch.Tensor) else np.asarray(v)) for k, v in arrs.items()})
    print(f"{name}: {os.path.getsize(path) / 1024:.0f} KiB")


def _randn(seed, *shape):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g)


def build_ref_pixel_decoder(ref, ch):
    ishape = {k: ref.ShapeSpec(channels=c, stride=s) for (k, c), s in zip(ch.items(), [4, 8, 16, 32])}
    pd = ref.pixdec.MSDeformAttnPixelDecoder(
        ishape, transformer_dropout=0.1, transformer_nheads=8, transformer_dim_feedforward=1024,
        transformer_enc_layers=6, conv_dim=256, mask_dim=256, norm="GN",
        transformer_in_features=["res3", "res4", "res5"], common_stride=4)
    pd.eval()
    fill.fill_module(pd, "sem_seg_head.pixel_decoder.")
    return pd


def build_ref_decoder(ref, is_train=False):
    dec = ref.dec.ContrastiveMultiScaleMaskedTransformerDecoder(
        256, True, num_classes=19, hidden_dim=256, num_queries=150, nheads=8, dropout=0.1,
        dim_feedforward=2048, enc_layers=0, is_train=is_train, dec_layers=9, class_dec_layers=2,
        pre_norm=False, mask_dim=256, enforce_input_project=False, use_task_norm=True)
    dec.eval()
    fill.fill_module(dec, "sem_seg_head.predictor.")
    return dec


def build_ref_swin(ref, cfg: T.SwinCfg):
    m = ref.swin.SwinTransformer(embed_dim=cfg.embed_dim, depths=list(cfg.depths), num_heads=list(cfg.num_heads),
                                 window_size=cfg.window_size, drop_path_rate=0.3)
    m.eval()
    fill.fill_module(m, "backbone.")
    return m


def decoder_contrastive(ref):
    """F11: the reference decoder constructed with is_train=True (oneformer_transformer_decoder.py:273, 477-480) on F7's inputs.
    Loss = mean(contrastive_logits^2) + 0.1 * synthetic loss; stores the contrastive tensor and three gradients."""
    f7 = np.load(os.path.join(OUT, "transformer_decoder.npz"))
    mf = torch.from_numpy(f7["mask_features"])
    ms = [torch.from_numpy(f7[f"ms{i}"]) for i in range(3)]
    tasks = torch.from_numpy(f7["tasks"]).requires_grad_()
    dec = build_ref_decoder(ref, is_train=True)
    o = dec(ms, mf, tasks)
    cl = o["contrastive_logits"]                       # (B, Q, E) after the reference's permute
    loss = cl.square().mean() + 0.1 * T.synthetic_loss(o)
    loss.backward()
    named = dict(dec.named_parameters())
    _save("decoder_contrastive", contrastive_logits=cl, loss=loss.detach(), grad_tasks=tasks.grad,
          grad_query_embed=named["query_embed.weight"].grad,
          grad_class_norm_weight=named["class_transformer.decoder.norm.weight"].grad,
          grad_ct_l1_linear2_bias=named["class_transformer.decoder.layers.1.linear2.bias"].grad)


def swin_ape(ref):
    """F12: the reference SwinTransformer built with ape=True (absolute position embedding at pretrain_img_size 64, resized bicubically
    to the token grid of a 64 x 96 input, swin.py:566-578, 656-661) and drop_rate = attn_drop_rate = 0.1 (identity in eval mode),
    C 64, depths 1-1-1-1, ws 7; forward and the embedding's gradient under a fixed linear loss."""
    m = ref.swin.SwinTransformer(pretrain_img_size=64, embed_dim=64, depths=[1, 1, 1, 1], num_heads=[2, 4, 8, 16], window_size=7,
                                 drop_rate=0.1, attn_drop_rate=0.1, drop_path_rate=0.0, ape=True)
    m.eval()
    fill.fill_module(m, "backbone.")
    img = _randn(12, 1, 3, 64, 96)
    o = m(img)
    wts = {k: _randn(40 + i, *o[k].shape) for i, k in enumerate(sorted(o))}
    sum((o[k] * wts[k]).sum() for k in o).backward()
    _save("swin_ape", img=img, grad_ape=m.absolute_pos_embed.grad, ape_shape=np.array(m.absolute_pos_embed.shape),
          **{k: v for k, v in o.items()}, **{"w_" + k: v for k, v in wts.items()})


def main():
    warnings.filterwarnings("ignore")
    assert ref_loader.available(), "needs /root/reference"
    os.makedirs(OUT, exist_ok=True)
    ref = ref_loader.load()
    torch.manual_seed(0)
    manifest = {}
    only = set(sys.argv[1:])
    if only:
        for name in only:
            {"decoder_contrastive": decoder_contrastive, "swin_ape": swin_ape}[name](ref)
        return

    # F1 / F2: block pairs through the reference BasicLayer (builds the shift mask itself)
    for tag, C, nH, ws, H, W, seed in (("swin_pair_ws7", 96, 3, 7, 24, 40, 1), ("swin_pair_ws12", 192, 6, 12, 20, 30, 2)):
        layer = ref.swin.BasicLayer(dim=C, depth=2, num_heads=nH, window_size=ws, drop_path=[0.0, 0.1])
        layer.eval()
        fill.fill_module(layer, "backbone.layers.0.")
        x = _randn(seed, 1, H * W, C)
        with torch.no_grad():
            # also record the output after the first (un-shifted) block
            for blk in layer.blocks:
                blk.H, blk.W = H, W
            y0 = layer.blocks[0](x, None)
            y = layer(x, H, W)[0]
        _save(tag, x=x, y_block0=y0, y=y, meta=np.array([C, nH, ws, H, W]))
        manifest[tag] = dict(C=C, nH=nH, ws=ws, H=H, W=W)

    # F3: full Swin-T
    cfg = T.SWIN_T
    m = build_ref_swin(ref, cfg)
    img = _randn(3, 1, 3, 96, 160)
    with torch.no_grad():
        o = m(img)
    _save("swin_t_96x160", img=img, **o)

    # F4: patch merging on odd H, W
    pm = ref.swin.PatchMerging(32)
    pm.eval()
    fill.fill_module(pm, "backbone.layers.0.downsample.")
    x = _randn(4, 2, 7 * 9, 32)
    with torch.no_grad():
        y = pm(x, 7, 9)
    _save("patch_merging", x=x, y=y, meta=np.array([32, 7, 9]))

    # F5: deformable-attention core, forward + the three gradients (autograd of the reference's
    # grid_sample formulation, ops/functions/ms_deform_attn_func.py:55-75)
    shapes = [(3, 4), (6, 8), (12, 16)]
    S = sum(h * w for h, w in shapes)
    B, Lq, M, D, L, P = 2, 40, 8, 32, 3, 4
    value = _randn(5, B, S, M, D).requires_grad_()
    g = torch.Generator().manual_seed(6)
    loc = (torch.rand(B, Lq, M, L, P, 2, generator=g) * 1.3 - 0.15).requires_grad_()  # some samples fall outside
    w = torch.rand(B, Lq, M, L, P, generator=g)
    w = (w / w.sum((-1, -2), keepdim=True)).requires_grad_()
    out = ref.msda_func.ms_deform_attn_core_pytorch(value, shapes, loc, w)
    go = _randn(7, *out.shape)
    out.backward(go)
    _save("msdeform_core", value=value, loc=loc, w=w, out=out, grad_out=go, grad_value=value.grad,
          grad_loc=loc.grad, grad_w=w.grad, shapes=np.array(shapes))

    # F6 / F7: pixel decoder and transformer decoder
    ch = {"res2": 96, "res3": 192, "res4": 384, "res5": 768}
    pd = build_ref_pixel_decoder(ref, ch)
    dec = build_ref_decoder(ref)
    H, W = 64, 96
    feats = {k: _randn(10 + i, 1, c, H // s, W // s) for i, ((k, c), s) in enumerate(zip(ch.items(), [4, 8, 16, 32]))}
    tasks = _randn(20, 1, 256)
    with torch.no_grad():
        mf, enc0, ms = pd.forward_features(feats)
        o = dec(ms, mf, tasks)
        ams = T.transformer_decoder(ms, mf, tasks, {"sem_seg_head.predictor." + k: v for k, v in dec.state_dict().items()},
                                    T.HeadCfg())["attn_masks"]
    _save("pixel_decoder", **feats, mask_features=mf, ms0=ms[0], ms1=ms[1], ms2=ms[2])
    _save("transformer_decoder", mask_features=mf, ms0=ms[0], ms1=ms[1], ms2=ms[2], tasks=tasks,
          pred_logits=o["pred_logits"], pred_masks=o["pred_masks"],
          **{f"aux{i}_logits": a["pred_logits"] for i, a in enumerate(o["aux_outputs"])},
          **{f"aux{i}_masks": a["pred_masks"].half() for i, a in enumerate(o["aux_outputs"])},
          **{f"attn_mask{i}": np.packbits(_np(a)) for i, a in enumerate(ams)})

    # F8: sine position embedding
    pe = ref.pe.PositionEmbeddingSine(128, normalize=True)
    _save("pos_embed_sine", pos=pe(torch.zeros(1, 4, 5, 7)))

    # F9: tokenizer ids (reference tokenizer with ftfy.fix_text stubbed to identity: ASCII prompts)
    ids = {}
    try:
        sys.modules.setdefault("ftfy", types.SimpleNamespace(fix_text=lambda s: s))
        tok = ref_loader._load("model.data.tokenizer", "data/tokenizer.py")
        tk = tok.Tokenize(tok.SimpleTokenizer(), max_seq_len=77)
        for name in T.TASK_TOKEN_IDS:
            ids[name] = tk(name).numpy()
            assert (ids[name] == T.task_tokens(name).numpy()).all(), name
        print("tokenizer ids verified against the reference tokenizer")
    except Exception as e:  # vocabulary file unreadable etc.: keep the surveyed constants
        print("tokenizer not importable:", repr(e))
        ids = {name: T.task_tokens(name).numpy() for name in T.TASK_TOKEN_IDS}
    tm = ref.dec.MLP(77, 256, 256, 2)
    fill.fill_module(tm, "task_mlp.")
    with torch.no_grad():
        emb = tm(torch.from_numpy(np.stack(list(ids.values()))).float())
    _save("task_tokens", names=np.array(list(ids.keys())), ids=np.stack(list(ids.values())), task_mlp_out=emb)

    # F10: small full model forward + backward through the reference modules
    scfg = T.SwinCfg(64, (2, 2, 2, 2), (2, 4, 8, 16), 7)
    sw = build_ref_swin(ref, scfg)
    ch = {f"res{i + 2}": 64 * 2 ** i for i in range(4)}
    pd = build_ref_pixel_decoder(ref, ch)
    dec = build_ref_decoder(ref)
    g = torch.Generator().manual_seed(30)
    imgs = [torch.randint(0, 256, (3, 64, 96), generator=g).float(), torch.randint(0, 256, (3, 64, 96), generator=g).float()]
    mcfg = T.ModelCfg(swin=scfg)
    x = T.preprocess(imgs, mcfg)
    tk_sd = {"task_mlp." + k: v for k, v in tm.state_dict().items()}
    tasks = T.task_embedding(["The task is panoptic", "The task is semantic"], tk_sd, mcfg)
    feats = sw(x)
    mf, _, ms = pd.forward_features(feats)
    o = dec(ms, mf, tasks)
    loss = T.synthetic_loss(o)
    loss.backward()
    named = dict(("backbone." + k, v) for k, v in sw.named_parameters())
    named.update(("sem_seg_head.pixel_decoder." + k, v) for k, v in pd.named_parameters())
    named.update(("sem_seg_head.predictor." + k, v) for k, v in dec.named_parameters())
    pick = ["backbone.patch_embed.proj.weight", "backbone.layers.0.blocks.0.attn.qkv.weight",
            "backbone.layers.0.blocks.1.attn.relative_position_bias_table", "backbone.layers.0.blocks.1.attn.qkv.bias",
            "backbone.layers.3.blocks.1.mlp.fc2.weight", "backbone.layers.1.downsample.reduction.weight",
            "backbone.norm2.weight",
            "sem_seg_head.pixel_decoder.transformer.encoder.layers.0.self_attn.sampling_offsets.weight",
            "sem_seg_head.pixel_decoder.transformer.encoder.layers.5.self_attn.value_proj.weight",
            "sem_seg_head.pixel_decoder.layer_1.weight",
            "sem_seg_head.predictor.query_embed.weight", "sem_seg_head.predictor.class_embed.weight",
            "sem_seg_head.predictor.class_transformer.decoder.layers.0.multihead_attn.in_proj_weight",
            "sem_seg_head.predictor.transformer_cross_attention_layers.4.multihead_attn.in_proj_weight",
            "sem_seg_head.predictor.mask_embed.layers.2.weight"]
    sq = sum(float(p.grad.double().square().sum()) for p in named.values() if p.grad is not None)
    # big gradients are stored as a strided sample of the flattened tensor plus their L2 norm
    grads = {}
    for i, n in enumerate(pick):
        gflat = named[n].grad.reshape(-1)
        stride = max(1, -(-gflat.numel() // 16384))
        grads[f"grad{i}"] = gflat[::stride].clone()
        grads[f"gradnorm{i}"] = gflat.double().norm().float()
        grads[f"gradstride{i}"] = np.array(stride)
    _save("model_fwd_bwd", img0=imgs[0].byte(), img1=imgs[1].byte(), loss=loss.detach(), pred_logits=o["pred_logits"],
          pred_masks=o["pred_masks"], grad_names=np.array(pick), grad_norm=np.array(sq ** 0.5), **grads)
    decoder_contrastive(ref)
    swin_ape(ref)
    with open(os.path.join(OUT, "MANIFEST.json"), "w") as f:
        json.dump({"generator": "python -m oracle.make_golden", "torch": torch.__version__,
                   "note": "inputs + outputs of the reference's modules with name-hashed weights (oracle/fill.py)"}, f, indent=1)


if __name__ == "__main__":
    main()
