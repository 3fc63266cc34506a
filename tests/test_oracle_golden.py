"""The oracle (oracle/torch_ref.py) against the committed fixtures the reference produced.

CPU only.  Weights are regenerated from parameter names (oracle/fill.py); the fixtures hold
inputs and the reference modules' outputs (oracle/make_golden.py).
"""
import numpy as np
import pytest
import torch

from conftest import load_golden
from oracle import fill
from oracle import torch_ref as T

TOL = dict(atol=2e-5, rtol=1e-5)


def _sd(shapes):
    return fill.state_dict_for(shapes)


@pytest.mark.parametrize("tag", ["swin_pair_ws7", "swin_pair_ws12"])
def test_swin_block_pair(tag):
    g = load_golden(tag)
    C, nH, ws, H, W = [int(v) for v in g["meta"]]
    cfg = T.SwinCfg(C, (2,), (nH,), ws)
    sd = _sd({k: s for k, s in T.swin_param_shapes(cfg).items() if ".layers.0.blocks." in k})
    with torch.no_grad():
        y0 = T.swin_block(g["x"], sd, "backbone.layers.0.blocks.0", H, W, ws, 0, nH)
        y1 = T.swin_block(y0, sd, "backbone.layers.0.blocks.1", H, W, ws, ws // 2, nH)
    torch.testing.assert_close(y0, g["y_block0"], **TOL)
    torch.testing.assert_close(y1, g["y"], **TOL)


def test_swin_t_backbone():
    g = load_golden("swin_t_96x160")
    sd = _sd(T.swin_param_shapes(T.SWIN_T))
    with torch.no_grad():
        o = T.swin_backbone(g["img"], sd, T.SWIN_T)
    for k in ("res2", "res3", "res4", "res5"):
        torch.testing.assert_close(o[k], g[k], **TOL)


def test_patch_merging_odd():
    g = load_golden("patch_merging")
    C, H, W = [int(v) for v in g["meta"]]
    p = "backbone.layers.0.downsample"
    sd = _sd({p + ".reduction.weight": (2 * C, 4 * C), p + ".norm.weight": (4 * C,), p + ".norm.bias": (4 * C,)})
    torch.testing.assert_close(T.patch_merging(g["x"], sd, p, H, W), g["y"], **TOL)


def test_msdeform_core_forward_and_grads():
    g = load_golden("msdeform_core")
    shapes = [tuple(int(v) for v in r) for r in g["shapes"]]
    value, loc, w = (g[k].clone().requires_grad_() for k in ("value", "loc", "w"))
    out = T.ms_deform_attn_core(value, shapes, loc, w)
    torch.testing.assert_close(out, g["out"], **TOL)
    out.backward(g["grad_out"])
    torch.testing.assert_close(value.grad, g["grad_value"], **TOL)
    torch.testing.assert_close(w.grad, g["grad_w"], **TOL)
    torch.testing.assert_close(loc.grad, g["grad_loc"], atol=2e-4, rtol=1e-4)


def test_pixel_decoder():
    g = load_golden("pixel_decoder")
    ch = {k: g[k].shape[1] for k in ("res2", "res3", "res4", "res5")}
    sd = _sd({k: s for k, s in T.head_param_shapes(T.HeadCfg(), ch).items() if "pixel_decoder" in k})
    with torch.no_grad():
        mf, _, ms = T.pixel_decoder({k: g[k] for k in ch}, sd, T.HeadCfg())
    torch.testing.assert_close(mf, g["mask_features"], atol=1e-4, rtol=1e-4)
    for i in range(3):
        torch.testing.assert_close(ms[i], g[f"ms{i}"], atol=1e-4, rtol=1e-4)


def test_transformer_decoder():
    g = load_golden("transformer_decoder")
    ch = {"res2": 96, "res3": 192, "res4": 384, "res5": 768}
    sd = _sd({k: s for k, s in T.head_param_shapes(T.HeadCfg(), ch).items() if "predictor" in k})
    with torch.no_grad():
        o = T.transformer_decoder([g["ms0"], g["ms1"], g["ms2"]], g["mask_features"], g["tasks"], sd, T.HeadCfg())
    torch.testing.assert_close(o["pred_logits"], g["pred_logits"], atol=1e-4, rtol=1e-4)
    torch.testing.assert_close(o["pred_masks"], g["pred_masks"], atol=1e-3, rtol=1e-4)
    for i, a in enumerate(o["aux_outputs"]):
        torch.testing.assert_close(a["pred_logits"], g[f"aux{i}_logits"], atol=1e-4, rtol=1e-4)
        torch.testing.assert_close(a["pred_masks"], g[f"aux{i}_masks"].float(), atol=5e-2, rtol=2e-3)
    for i, m in enumerate(o["attn_masks"]):
        bits = np.unpackbits(g[f"attn_mask{i}"].numpy())[: m.numel()].reshape(m.shape)
        assert (m.numpy() == bits.astype(bool)).mean() > 0.9999


def test_pos_embed_sine():
    g = load_golden("pos_embed_sine")
    torch.testing.assert_close(T.position_embedding_sine(1, 5, 7, 128), g["pos"], **TOL)


def test_task_tokens():
    g = load_golden("task_tokens")
    for name, ids in zip(g["names"], g["ids"]):
        assert (T.task_tokens(str(name)).numpy() == ids.numpy()).all()
    sd = _sd({"task_mlp.layers.0.weight": (256, 77), "task_mlp.layers.0.bias": (256,),
              "task_mlp.layers.1.weight": (256, 256), "task_mlp.layers.1.bias": (256,)})
    emb = T.task_embedding([str(n) for n in g["names"]], sd, T.ModelCfg())
    torch.testing.assert_close(emb, g["task_mlp_out"], rtol=1e-5, atol=1e-2)  # inputs are token ids ~5e4


def test_model_forward_backward():
    g = load_golden("model_fwd_bwd")
    cfg = T.ModelCfg(swin=T.SwinCfg(64, (2, 2, 2, 2), (2, 4, 8, 16), 7))
    sd = {k: v.requires_grad_() for k, v in _sd(T.model_param_shapes(cfg)).items()}
    batch = [{"left_image": g["img0"].float(), "task": "The task is panoptic"},
             {"left_image": g["img1"].float(), "task": "The task is semantic"}]
    out = T.oneformer_forward(batch, sd, cfg, upsample=False)
    loss = T.synthetic_loss(out)
    torch.testing.assert_close(loss, g["loss"], atol=1e-4, rtol=1e-4)
    torch.testing.assert_close(out["pred_logits"], g["pred_logits"], atol=2e-4, rtol=1e-4)
    loss.backward()
    for i, n in enumerate(g["grad_names"]):
        gr = sd[str(n)].grad.reshape(-1)
        torch.testing.assert_close(gr.double().norm().float(), g[f"gradnorm{i}"], atol=1e-6, rtol=2e-3)
        ref = g[f"grad{i}"]
        got = gr[:: int(g[f"gradstride{i}"])]
        assert (got - ref).abs().max() <= 2e-3 * ref.abs().max() + 1e-7, n


def test_transformer_decoder_is_train_contrastive():
    """`contrastive_logits` of a decoder built with is_train=True is the PRE-loop query tensor
    (oneformer_transformer_decoder.py:440, 477-480); value and gradients against the reference's own run (fixture F11)."""
    g, c = load_golden("transformer_decoder"), load_golden("decoder_contrastive")
    ch = {"res2": 96, "res3": 192, "res4": 384, "res5": 768}
    sd = {k: v.requires_grad_() for k, v in _sd({k: s for k, s in T.head_param_shapes(T.HeadCfg(), ch).items() if "predictor" in k}).items()}
    tasks = g["tasks"].clone().requires_grad_()
    o = T.transformer_decoder([g["ms0"], g["ms1"], g["ms2"]], g["mask_features"], tasks, sd, T.HeadCfg(), is_train=True)
    cl = o["contrastive_logits"]
    torch.testing.assert_close(cl, c["contrastive_logits"], atol=1e-4, rtol=1e-4)
    loss = cl.square().mean() + 0.1 * T.synthetic_loss(o)
    torch.testing.assert_close(loss, c["loss"], atol=1e-4, rtol=1e-4)
    loss.backward()
    p = "sem_seg_head.predictor."
    for got, key in ((tasks.grad, "grad_tasks"), (sd[p + "query_embed.weight"].grad, "grad_query_embed"),
                     (sd[p + "class_transformer.decoder.norm.weight"].grad, "grad_class_norm_weight"),
                     (sd[p + "class_transformer.decoder.layers.1.linear2.bias"].grad, "grad_ct_l1_linear2_bias")):
        ref = c[key]
        assert (got - ref).abs().max() <= 2e-3 * ref.abs().max() + 1e-7, key
