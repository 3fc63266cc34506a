"""The fp32 "exact" arithmetic mode (csrc/exact.hip, `uenc.ops.set_exact`) against the reference's fixtures and the oracle.

SURVEY.md §8(c) states two tolerance classes: the product's bf16-MFMA mode (per-output tolerances, tests/test_model_gpu.py)
and a HIP fp32 mode that must agree with the fp32 reference to <= 1e-4 -- the mode in which a FREE-RUNNING decoder (its
boolean attention masks are thresholded intermediate predictions) can be compared at all.  Tolerances here, relative L2 per
tensor unless stated: kernels 2e-5; module fixtures 1e-4; free-running decoder / full model logits and mask logits 1e-3 with
>= 99.9 % mask-sign agreement; parameter gradients rel 2e-3 (float atomics order).  The bf16-mode figures of the same
free-running runs are recorded next to the exact ones in the parity file (tests/conftest.py::record_parity).
"""
import os

import numpy as np
import pytest
import torch

from conftest import load_golden, mask_band_figures, record_parity

pytestmark = pytest.mark.gpu


def rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))


@pytest.fixture(scope="module")
def U():
    import model  # noqa: F401
    import uenc
    return uenc


@pytest.fixture()
def exact(U):
    from uenc import ops
    ops.set_exact(True)
    yield ops
    ops.set_exact(False)


def _fill(module, prefix=""):
    from oracle import fill
    fill.fill_module(module, prefix)
    return module


# ---------------------------------------------------------------------------------------------------------------------
# kernels
# ---------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("M,N,K", [(300, 200, 48), (129, 136, 80), (1024, 768, 192), (150, 20 + 4, 256), (64, 256, 2048)])
def test_gemm_nt_f32_epilogues(exact, M, N, K):
    from uenc import kernels as Kk
    g = torch.Generator().manual_seed(M + N + K)
    a = torch.randn(M, K, generator=g).cuda()
    w = torch.randn(N, K, generator=g).cuda()
    bias = torch.randn(N, generator=g).cuda()
    aux = torch.randn(M, N, generator=g).cuda()
    ref0 = (a.double() @ w.double().t() + bias.double())
    worst = 0.0
    for epi, alpha in ((Kk.EPI_NONE, 1.0), (Kk.EPI_NONE, 0.5), (Kk.EPI_GELU, 1.0), (Kk.EPI_RELU, 1.0), (Kk.EPI_RESIDUAL, 2.0),
                       (Kk.EPI_MUL_DGELU, 1.0), (Kk.EPI_MUL_DRELU, 1.0)):
        pre = torch.empty(M, N, device="cuda") if epi == Kk.EPI_GELU else None
        out = Kk.gemm_nt(a, w, bias=bias, epilogue=epi, aux=aux if epi >= Kk.EPI_RESIDUAL else None, aux_out=pre, alpha=alpha)
        v = ref0 * alpha
        if epi == Kk.EPI_GELU:
            assert rel(pre, v) < 2e-6
            v = torch.nn.functional.gelu(v)
        elif epi == Kk.EPI_RELU:
            v = v.relu()
        elif epi == Kk.EPI_RESIDUAL:
            v = v + aux.double()
        elif epi == Kk.EPI_MUL_DGELU:
            x = aux.double().requires_grad_()
            torch.nn.functional.gelu(x).sum().backward()
            v = v * x.grad
        elif epi == Kk.EPI_MUL_DRELU:
            v = v * (aux.double() > 0)
        assert out.dtype == torch.float32
        worst = max(worst, rel(out, v))
    record_parity(f"exact/gemm_nt_f32[{M}x{N}x{K}]", worst_rel=worst)
    assert worst < 2e-6
    # accumulate
    c = torch.randn(M, N, generator=g).cuda()
    c0 = c.clone()
    Kk.gemm_nt(a, w, out=c, accumulate=True)
    assert rel(c, c0.double() + a.double() @ w.double().t()) < 2e-6


@pytest.mark.parametrize("M,N,K", [(1000, 96, 48), (4096, 256, 1024), (300, 20 + 4, 256)])
def test_gemm_tn_f32(exact, M, N, K):
    from uenc import kernels as Kk
    g = torch.Generator().manual_seed(M + N)
    dy = torch.randn(M, N, generator=g).cuda()
    x = torch.randn(M, K, generator=g).cuda()
    dw = torch.randn(N, K, generator=g).cuda()
    db = torch.randn(N, generator=g).cuda()
    dw0, db0 = dw.clone(), db.clone()
    Kk.gemm_tn(dy, x, dw, db)
    r1, r2 = rel(dw, dw0.double() + dy.double().t() @ x.double()), rel(db, db0.double() + dy.double().sum(0))
    record_parity(f"exact/gemm_tn_f32[{M}x{N}x{K}]", dw_rel=r1, db_rel=r2)
    assert r1 < 3e-6 and r2 < 3e-6


@pytest.mark.parametrize("B,Lq,S,masked", [(2, 150, 96, True), (1, 149, 2048 + 37, False), (2, 7, 513, True)])
def test_mha_f32_fwd_bwd(exact, B, Lq, S, masked):
    from uenc.attention import mha
    nH, E = 8, 256
    g = torch.Generator().manual_seed(S)
    q, k, v = (torch.randn(B, n, E, generator=g).cuda().requires_grad_() for n in (Lq, S, S))
    mask = None
    if masked:
        mask = torch.rand(B, Lq, S, generator=g) < 0.6
        mask[:, :, 0] = False                       # no fully blocked row
        mask = mask.cuda()
    out = mha(q, k, v, nH, mask)
    go = torch.randn(B, Lq, E, generator=g).cuda()
    out.backward(go)
    qd, kd, vd = (t.detach().double().cpu().requires_grad_() for t in (q, k, v))
    qh, kh, vh = (t.view(B, -1, nH, 32).transpose(1, 2) for t in (qd, kd, vd))
    s = qh @ kh.transpose(-1, -2) / 32 ** 0.5
    if mask is not None:
        s = s.masked_fill(mask.cpu()[:, None], float("-inf"))
    ref = (s.softmax(-1) @ vh).transpose(1, 2).reshape(B, Lq, E)
    ref.backward(go.double().cpu())
    figs = dict(out=rel(out, ref), dq=rel(q.grad, qd.grad), dk=rel(k.grad, kd.grad), dv=rel(v.grad, vd.grad))
    record_parity(f"exact/mha_f32[B{B} Lq{Lq} S{S} mask{int(masked)}]", **figs)
    assert all(x < 1e-5 for x in figs.values()), figs


# ---------------------------------------------------------------------------------------------------------------------
# modules against the reference's fixtures, exact mode
# ---------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("tag", ["swin_pair_ws7", "swin_pair_ws12"])
def test_swin_block_pair_exact(exact, tag):
    """Window attention (padding slots, shift, -100 mask), GELU MLP, LayerNorm -- forward vs the reference fixture, and the
    backward against autograd of the fp32 oracle."""
    from oracle import fill, torch_ref as T
    from uenc.modeling.backbone.swin import BasicLayer
    g = load_golden(tag)
    C, nH, ws, H, W = [int(v) for v in g["meta"]]
    layer = _fill(BasicLayer(dim=C, depth=2, num_heads=nH, window_size=ws), "backbone.layers.0.").cuda().eval()
    x = g["x"].cuda().requires_grad_()
    y = layer(x, H, W)[0]
    r = rel(y, g["y"])
    go = torch.randn(y.shape, generator=torch.Generator().manual_seed(1))
    y.backward(go.cuda())
    sd = {k: v.requires_grad_() for k, v in fill.state_dict_for(
        {k: s for k, s in T.swin_param_shapes(T.SwinCfg(C, (2,), (nH,), ws)).items() if ".layers.0.blocks." in k}).items()}
    xo = g["x"].clone().requires_grad_()
    y0 = T.swin_block(xo, sd, "backbone.layers.0.blocks.0", H, W, ws, 0, nH)
    y1 = T.swin_block(y0, sd, "backbone.layers.0.blocks.1", H, W, ws, ws // 2, nH)
    y1.backward(go)
    named = dict(layer.named_parameters())
    figs = {"y": r, "dx": rel(x.grad, xo.grad)}
    for n in ("blocks.1.attn.qkv.weight", "blocks.1.attn.qkv.bias", "blocks.1.attn.relative_position_bias_table", "blocks.0.mlp.fc1.weight",
              "blocks.0.norm1.weight", "blocks.1.attn.proj.bias"):
        figs["d_" + n] = rel(named[n].grad, sd["backbone.layers.0." + n].grad)
    record_parity(f"exact/{tag}", **figs)
    assert r < 1e-5, r
    assert all(v < 2e-4 for v in figs.values()), figs


def test_swin_t_backbone_exact(exact):
    from uenc.modeling.backbone.swin import SwinTransformer
    g = load_golden("swin_t_96x160")
    m = _fill(SwinTransformer(embed_dim=96, depths=[2, 2, 6, 2], num_heads=[3, 6, 12, 24], window_size=7), "backbone.").cuda()
    m.eval()
    with torch.no_grad():
        o = m(g["img"].cuda())
    figs = {k: rel(o[k], g[k]) for k in ("res2", "res3", "res4", "res5")}
    record_parity("exact/swin_t_96x160", **figs)
    assert all(v < 1e-4 for v in figs.values()), figs


def _head_modules(ch):
    from uenc.d2 import ShapeSpec
    from uenc.modeling.pixel_decoder.msdeformattn import MSDeformAttnPixelDecoder
    from uenc.modeling.transformer_decoder.oneformer_transformer_decoder import ContrastiveMultiScaleMaskedTransformerDecoder
    ishape = {k: ShapeSpec(channels=c, stride=s) for (k, c), s in zip(ch.items(), [4, 8, 16, 32])}
    pd = MSDeformAttnPixelDecoder(ishape, transformer_dropout=0.1, transformer_nheads=8, transformer_dim_feedforward=1024,
                                  transformer_enc_layers=6, conv_dim=256, mask_dim=256, norm="GN",
                                  transformer_in_features=["res3", "res4", "res5"], common_stride=4)
    dec = ContrastiveMultiScaleMaskedTransformerDecoder(
        256, True, num_classes=19, hidden_dim=256, num_queries=150, nheads=8, dropout=0.1, dim_feedforward=2048, enc_layers=0,
        is_train=False, dec_layers=9, class_dec_layers=2, pre_norm=False, mask_dim=256, enforce_input_project=False,
        use_task_norm=True)
    return (_fill(pd, "sem_seg_head.pixel_decoder.").cuda().eval(), _fill(dec, "sem_seg_head.predictor.").cuda().eval())


def test_pixel_decoder_exact(exact):
    g = load_golden("pixel_decoder")
    ch = {k: g[k].shape[1] for k in ("res2", "res3", "res4", "res5")}
    pd, _ = _head_modules(ch)
    with torch.no_grad():
        mf, _, ms = pd.forward_features({k: g[k].cuda() for k in ch})
    figs = {"mask_features": rel(mf, g["mask_features"]), **{f"ms{i}": rel(ms[i], g[f"ms{i}"]) for i in range(3)}}
    record_parity("exact/pixel_decoder", **figs)
    assert all(v < 1e-4 for v in figs.values()), figs


def _decoder_figures(o, g):
    figs = {"pred_logits": rel(o["pred_logits"], g["pred_logits"]), "pred_masks": rel(o["pred_masks"], g["pred_masks"])}
    figs["mask_sign_agreement"] = float(((o["pred_masks"].cpu() > 0) == (g["pred_masks"] > 0)).float().mean())
    for i, a in enumerate(o["aux_outputs"]):
        figs[f"aux{i}_logits"] = rel(a["pred_logits"], g[f"aux{i}_logits"])
    figs["mask_band"] = mask_band_figures(o["pred_masks"], g["pred_masks"])
    return figs


def test_transformer_decoder_free_running_exact(U):
    """The FREE-RUNNING decoder (no forced masks): exact mode meets the fp32 tolerance; the bf16 mode's figures on the very
    same inputs are recorded beside it."""
    from uenc import ops
    g = load_golden("transformer_decoder")
    ch = {"res2": 96, "res3": 192, "res4": 384, "res5": 768}
    feats = [g["ms0"].cuda(), g["ms1"].cuda(), g["ms2"].cuda()]
    out = {}
    for mode in ("bf16", "exact"):
        ops.set_exact(mode == "exact")
        try:
            _, dec = _head_modules(ch)
            with torch.no_grad():
                out[mode] = _decoder_figures(dec(feats, g["mask_features"].cuda(), g["tasks"].cuda()), g)
        finally:
            ops.set_exact(False)
    record_parity("free_running/transformer_decoder", exact=out["exact"], bf16=out["bf16"])
    e = out["exact"]
    assert e["pred_logits"] < 1e-3 and e["pred_masks"] < 1e-3 and e["mask_sign_agreement"] >= 0.999, e


def _small_model(device="cuda"):
    from uenc.d2 import get_cfg, build_model
    from uenc.config import add_common_config, add_swin_config, add_uni_encoder_config
    cfg = get_cfg()
    add_common_config(cfg); add_swin_config(cfg); add_uni_encoder_config(cfg)
    cfg.merge_from_list([
        "MODEL.META_ARCHITECTURE", "OneFormer", "MODEL.BACKBONE.NAME", "D2SwinTransformer", "MODEL.SWIN.EMBED_DIM", 64,
        "MODEL.SWIN.DEPTHS", [2, 2, 2, 2], "MODEL.SWIN.NUM_HEADS", [2, 4, 8, 16], "MODEL.SEM_SEG_HEAD.NAME", "OneFormerHead",
        "MODEL.SEM_SEG_HEAD.PIXEL_DECODER_NAME", "MSDeformAttnPixelDecoder", "MODEL.SEM_SEG_HEAD.NUM_CLASSES", 19,
        "MODEL.SEM_SEG_HEAD.CONVS_DIM", 256, "MODEL.SEM_SEG_HEAD.IN_FEATURES", ["res2", "res3", "res4", "res5"],
        "MODEL.SEM_SEG_HEAD.TRANSFORMER_ENC_LAYERS", 6, "MODEL.ONE_FORMER.TRANSFORMER_IN_FEATURE", "multi_scale_pixel_decoder",
        "MODEL.ONE_FORMER.NUM_OBJECT_QUERIES", 150, "MODEL.ONE_FORMER.DEC_LAYERS", 10, "MODEL.IS_TRAIN", False,
        "MODEL.PIXEL_MEAN", [123.675, 116.280, 103.530], "MODEL.PIXEL_STD", [58.395, 57.120, 57.375], "MODEL.DEVICE", device])
    m = build_model(cfg)
    _fill(m)
    m.eval()
    return m


def test_full_model_free_running_forward_backward_exact(U):
    """Small full model, FREE-RUNNING (fixture F10 = the reference modules' own forward + backward): exact mode reproduces the
    loss, logits, masks and the 15 stored parameter gradients; the bf16 mode's free-running figures are recorded next to them."""
    from oracle import torch_ref as T
    from uenc import ops
    g = load_golden("model_fwd_bwd")
    batch = [{"left_image": g["img0"].float(), "task": "The task is panoptic", "type": "segmentation"},
             {"left_image": g["img1"].float(), "task": "The task is semantic", "type": "segmentation"}]
    res = {}
    for mode in ("bf16", "exact"):
        ops.set_exact(mode == "exact")
        try:
            model = _small_model()
            out, _ = model.forward_features(batch)
            loss = T.synthetic_loss(out)
            loss.backward()
            ops.flush_wgrads()
            named = dict(model.named_parameters())
            figs = {"loss_rel": abs(float(loss) - float(g["loss"])) / abs(float(g["loss"])),
                    "pred_logits": rel(out["pred_logits"], g["pred_logits"]), "pred_masks": rel(out["pred_masks"], g["pred_masks"]),
                    "mask_sign_agreement": float(((out["pred_masks"].detach().cpu() > 0) == (g["pred_masks"] > 0)).float().mean()),
                    "mask_band": mask_band_figures(out["pred_masks"], g["pred_masks"])}
            grads = {}
            for i, n in enumerate(g["grad_names"]):
                n = str(n)
                want, stride = g[f"grad{i}"], int(g[f"gradstride{i}"])
                gr = named[n].grad.reshape(-1).float().cpu()
                grads[n] = {"cos": float(torch.nn.functional.cosine_similarity(gr[::stride].double(), want.double(), dim=0)),
                            "norm_ratio": float(gr.double().norm()) / float(g[f"gradnorm{i}"]),
                            "rel": rel(gr[::stride], want)}
            figs["grads"] = grads
            res[mode] = figs
        finally:
            ops.set_exact(False)
    record_parity("free_running/small_full_model_fwd_bwd", exact=res["exact"], bf16=res["bf16"])
    e = res["exact"]
    assert e["loss_rel"] < 1e-4 and e["pred_logits"] < 1e-3 and e["pred_masks"] < 1e-3 and e["mask_sign_agreement"] >= 0.999, e
    bad = {n: v for n, v in e["grads"].items() if v["rel"] > 5e-3 or abs(v["norm_ratio"] - 1) > 2e-3}
    assert not bad, bad


def test_full_size_swin_l_free_running_exact(U):
    """BASELINE configs[2]'s model at its own size (Swin-L ws 12, one 1024 x 2048 image), FREE-RUNNING, against the fp32 oracle's
    forward on the host cores: exact mode <= 1e-3 on logits / mask logits and >= 99.9 % mask-sign agreement; bf16 mode recorded."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
    import bench
    from oracle import fill, torch_ref as T
    from uenc import ops
    from uenc.d2 import build_model
    g = torch.Generator().manual_seed(7)
    img = torch.randint(0, 256, (3, bench.H_IMG, bench.W_IMG), generator=g).float()
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    sd = fill.state_dict_for(T.model_param_shapes(T.ModelCfg(swin=T.SWIN_L)))
    with torch.no_grad():
        oref = T.oneformer_forward([{"left_image": img, "task": "The task is panoptic"}], sd, T.ModelCfg(swin=T.SWIN_L), upsample=False)
    batch = [{"left_image": img.cuda(), "task": "The task is panoptic", "type": "segmentation", "height": bench.H_IMG, "width": bench.W_IMG}]
    res = {}
    for mode in ("bf16", "exact"):
        ops.set_exact(mode == "exact")
        try:
            model = build_model(bench.make_cfg("cuda"))
            _fill(model)
            model.eval()
            with torch.no_grad():
                out, _ = model.forward_features(batch)
            res[mode] = {"pred_logits": rel(out["pred_logits"], oref["pred_logits"]), "pred_masks": rel(out["pred_masks"], oref["pred_masks"]),
                         "mask_sign_agreement": float(((out["pred_masks"].cpu() > 0) == (oref["pred_masks"] > 0)).float().mean()),
                         "aux_logits": [rel(a["pred_logits"], b["pred_logits"]) for a, b in zip(out["aux_outputs"], oref["aux_outputs"])],
                         "mask_band": mask_band_figures(out["pred_masks"], oref["pred_masks"]),
                         "aux_mask_band": [mask_band_figures(a["pred_masks"], b["pred_masks"]) for a, b in zip(out["aux_outputs"], oref["aux_outputs"])]}
            del model, out
            torch.cuda.empty_cache()
        finally:
            ops.set_exact(False)
    record_parity("free_running/swin_l_1024x2048_one_image", exact=res["exact"], bf16=res["bf16"])
    e = res["exact"]
    assert e["pred_logits"] < 1e-3 and e["pred_masks"] < 1e-3 and e["mask_sign_agreement"] >= 0.999, e
    assert e["mask_band"]["flips_outside_band"] == 0 and e["mask_band"]["abs_err_max"] < 5e-2, e["mask_band"]      # the contract's abs unit, fp32 mode
    # Product (bf16) mode, measured in the contract's units (logits <= 2e-2 rel, masks <= 5e-2 abs, sign >= 99.9 %).
    # The logits bound holds; the mask bounds do NOT (round 3 measurement, profiles/r03_parity_error_sources.json): mean abs error 0.051,
    # sign agreement 99.72 %, 71 % of the flips inside the +-5e-2 band, 99.92 % agreement outside it.  The error is the accumulated
    # operand rounding of all three stages (backbone alone -> 99.80 %, + pixel decoder -> 99.86 %, all fp32 -> 99.9999 %); the
    # thresholded mask path alone in fp32 buys 0.06 points.  The asserts pin the measured level so that it cannot degrade silently.
    b = res["bf16"]
    assert b["pred_logits"] < 2e-2, b["pred_logits"]
    mb = b["mask_band"]
    assert mb["abs_err_mean"] < 7e-2 and mb["mask_sign_agreement"] > 0.996 and mb["sign_agreement_outside_band"] > 0.9985 and \
        mb["flips_inside_band_share"] > 0.6 and mb["max_ref_abs_at_a_flip"] < 0.5, mb
