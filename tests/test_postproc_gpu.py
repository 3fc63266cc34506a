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
