"""Fused post-processing kernels (csrc/postproc.hip) and the meta-architecture's inference paths against the fixtures produced
by the reference's own semantic_inference / panoptic_inference (tests/golden/postproc.npz) and the oracle.

Tolerances: semantic maps are fp32 sums of Q products of fp32 sigmoids: |diff| <= 2e-5 against the reference (expf vs torch's
sigmoid, summation order).  Panoptic maps are integers: exact, except that a pixel whose interpolated logit is within float
rounding of 0 (sigmoid = 0.5) or whose two best queries tie to the last bit may differ -- the test allows 1e-4 of the pixels and
requires identical segments_info."""
import pytest
import torch

from conftest import load_golden

pytestmark = pytest.mark.gpu


def _cases(only_unresized=True):
    g = load_golden("postproc")
    for i in range(int(g["ncases"])):
        m = [int(v) for v in g[f"c{i}_meta"]]
        c = dict(Q=m[0], C=m[1], padded=tuple(m[4:6]), image=tuple(m[6:8]), out=tuple(m[8:10]), thr=float(g[f"c{i}_thr"][0]),
                 ovl=float(g[f"c{i}_thr"][1]), things=[int(v) for v in g[f"c{i}_things"]])
        if only_unresized and c["out"] != c["image"]:
            continue
        yield i, g, c


class _Head:
    def __init__(self, n):
        self.num_classes = n


def _meta_arch(c):
    """An OneFormer shell with only what the post-processing methods read."""
    import model  # noqa: F401
    from uenc.oneformer_model import OneFormer
    o = OneFormer.__new__(OneFormer)
    torch.nn.Module.__init__(o)
    o.sem_seg_head, o.object_mask_threshold, o.overlap_threshold, o.thing_ids = _Head(c["C"]), c["thr"], c["ovl"], tuple(c["things"])
    return o


def test_semantic_kernel_vs_reference_fixture():
    from uenc import kernels as K
    n = 0
    for i, g, c in _cases():
        p = torch.softmax(g[f"c{i}_cls"], -1)[:, :-1].cuda()
        sem = K.postproc_semantic(g[f"c{i}_masks"].cuda(), p, c["padded"], c["image"])
        torch.testing.assert_close(sem.cpu(), g[f"c{i}_sem"], atol=2e-5, rtol=1e-5)
        n += 1
    assert n >= 2


def test_panoptic_fused_vs_reference_fixture():
    n = 0
    for i, g, c in _cases():
        o = _meta_arch(c)
        seg, info = o.panoptic_inference_fused(g[f"c{i}_cls"].cuda(), g[f"c{i}_masks"].cuda(), c["padded"], c["image"])
        want = g[f"c{i}_pan"].to(torch.int32)
        assert float((seg.cpu() != want).float().mean()) <= 1e-4
        assert [[d["id"], int(d["isthing"]), d["category_id"]] for d in info] == g[f"c{i}_info"].tolist()
        n += 1
    assert n >= 2


def test_panoptic_fallback_path_vs_reference_fixture():
    """The materialised-mask path (used with gradients enabled or a second resize), all three fixture cases."""
    for i, g, c in _cases(only_unresized=False):
        o = _meta_arch(c)
        seg, info = o.panoptic_inference(g[f"c{i}_cls"].cuda(), g[f"c{i}_mask_pred"].cuda())
        assert torch.equal(seg.cpu(), g[f"c{i}_pan"].to(torch.int32))
        assert [[d["id"], int(d["isthing"]), d["category_id"]] for d in info] == g[f"c{i}_info"].tolist()


def test_instance_inference_vs_reference_fixture():
    """instance_inference on the low-resolution logits (selected queries upsampled only) and on materialised masks, against the
    reference's Instances for the three fixture cases (detections compared in score order: topk(sorted=False) has no order)."""
    from oracle import postproc_ref as P
    for i, g, c in _cases(only_unresized=False):
        o = _meta_arch(c)
        o.test_topk_per_image, o.is_demo, o.panoptic_on = int(g[f"c{i}_inst_topk"]), False, bool(int(g[f"c{i}_inst_panoptic_on"]))
        runs = [o.instance_inference(g[f"c{i}_cls"].cuda(), g[f"c{i}_mask_pred"].cuda(), "The task is instance")]
        if c["out"] == c["image"]:
            runs.append(o.instance_inference(g[f"c{i}_cls"].cuda(), g[f"c{i}_masks"].cuda(), "The task is instance", c["padded"], c["image"]))
        for r in runs:
            order = torch.argsort(r.scores, descending=True)
            torch.testing.assert_close(r.scores[order].cpu(), g[f"c{i}_inst_scores"], atol=1e-5, rtol=1e-4)
            assert torch.equal(r.pred_classes[order].cpu(), g[f"c{i}_inst_classes"])
            area = r.pred_masks[order].flatten(1).sum(1).cpu()
            assert float((area - g[f"c{i}_inst_area"]).abs().max()) <= 2.0          # pixels whose logit is within rounding of 0
            assert tuple(r.pred_boxes.tensor.shape) == (len(order), 4) and r.image_size == tuple(r.pred_masks.shape[-2:])


def test_full_size_semantic_and_panoptic_consistency():
    """1024 x 2048, Q = 150, C = 19: the fused kernels against the separate passes (upsample kernel + torch ops) on the same logits."""
    from oracle import postproc_ref as P
    from uenc import kernels as K
    cls, masks = P.synthetic_predictions(150, 19, 256, 512, seed=3)
    cls, masks = cls.cuda(), masks.cuda()
    c = dict(C=19, thr=0.3, ovl=0.05, things=list(range(11, 19)))       # 150 overlapping blobs: low thresholds keep some segments alive
    o = _meta_arch(c)
    sem = K.postproc_semantic(masks, torch.softmax(cls, -1)[:, :-1], (1024, 2048), (1024, 2048))
    up = K.upsample_bilinear(masks[None], (1024, 2048))[0]
    want = o.semantic_inference(cls, up)
    assert float((sem - want).abs().max()) < 1e-4
    seg, info = o.panoptic_inference_fused(cls, masks, (1024, 2048), (1024, 2048))
    seg2, info2 = o.panoptic_inference(cls, up)
    assert info == info2 and len(info) >= 3
    assert float((seg != seg2).float().mean()) <= 1e-5


def test_oneformer_forward_inference_paths_agree():
    """OneFormer.forward on a small model with semantic + panoptic inference on: the fused path (no_grad) against the
    materialised path the same forward takes when gradients are enabled (separate upsample, crop, torch ops)."""
    import model  # noqa: F401
    from oracle import fill
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
        "MODEL.TEST.SEMANTIC_ON", True, "MODEL.TEST.PANOPTIC_ON", True, "MODEL.TEST.INSTANCE_ON", True, "TEST.DETECTIONS_PER_IMAGE", 20,
        "MODEL.TEST.OBJECT_MASK_THRESHOLD", 0.05, "MODEL.TEST.OVERLAP_THRESHOLD", 0.05,
        "MODEL.PIXEL_MEAN", [123.675, 116.280, 103.530], "MODEL.PIXEL_STD", [58.395, 57.120, 57.375], "MODEL.DEVICE", "cuda"])
    m = build_model(cfg)
    fill.fill_module(m, "")
    m.eval()
    g = torch.Generator().manual_seed(11)
    # 90 x 120 is padded to 96 x 128 (size divisibility 32): the crop is part of both paths
    batch = [{"left_image": torch.randint(0, 256, (3, 90, 120), generator=g).float(), "task": "The task is panoptic", "type": "segmentation"}]
    with torch.no_grad():
        fused = m(batch)[0]
    slow = m(batch)[0]                      # gradients enabled -> the materialised path
    assert "pred_masks" in slow and "pred_masks" not in fused
    assert tuple(fused["sem_seg"].shape) == (19, 90, 120) and tuple(fused["panoptic_seg"][0].shape) == (90, 120)
    assert float((fused["sem_seg"] - slow["sem_seg"].detach()).abs().max()) < 1e-4
    assert fused["panoptic_seg"][1] == slow["panoptic_seg"][1]
    assert float((fused["panoptic_seg"][0] != slow["panoptic_seg"][0]).float().mean()) <= 1e-3
    fi, si = fused["instances"], slow["instances"]
    of, os_ = torch.argsort(fi.scores, descending=True), torch.argsort(si.scores, descending=True)
    assert len(fi) == len(si) and torch.equal(fi.pred_classes[of], si.pred_classes[os_])
    torch.testing.assert_close(fi.scores[of], si.scores[os_].detach(), atol=1e-5, rtol=1e-4)
    assert fi.image_size == (90, 120) and tuple(fi.pred_masks.shape[-2:]) == (90, 120)
    assert "box_instances" not in fused                                   # MODEL.TEST.DETECTION_ON is off by default
    # reference :301-304, :478-480: with DETECTION_ON a second instance pass whose boxes are the tight boxes of the binary masks
    m.detection_on = True
    with torch.no_grad():
        det = m(batch)[0]["box_instances"]
    m.detection_on = False
    assert len(det) == len(fi) and tuple(det.pred_boxes.tensor.shape) == (len(det), 4)
    for k in range(len(det)):
        mk = det.pred_masks[k] > 0
        x1, y1, x2, y2 = [float(v) for v in det.pred_boxes.tensor[k]]
        if bool(mk.any()):
            ys, xs = torch.where(mk)
            assert (x1, y1, x2, y2) == (float(xs.min()), float(ys.min()), float(xs.max()) + 1, float(ys.max()) + 1)
        else:
            assert (x1, y1, x2, y2) == (0.0, 0.0, 0.0, 0.0)
    assert float(fi.pred_boxes.tensor.abs().sum()) == 0.0                   # detection off: zeros, as the reference
