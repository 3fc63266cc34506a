"""oracle/postproc_ref.py against the fixtures produced by the reference's own `semantic_inference` / `panoptic_inference`
(tests/golden/postproc.npz, made by oracle/make_postproc_golden.py), and live against the reference where it is present."""
import types

import pytest
import torch

from conftest import load_golden
from oracle import postproc_ref as P


def _cases():
    g = load_golden("postproc")
    for i in range(int(g["ncases"])):
        meta = [int(v) for v in g[f"c{i}_meta"]]
        yield i, g, dict(Q=meta[0], C=meta[1], h=meta[2], w=meta[3], padded=tuple(meta[4:6]), image=tuple(meta[6:8]), out=tuple(meta[8:10]),
                         thr=float(g[f"c{i}_thr"][0]), ovl=float(g[f"c{i}_thr"][1]), things=[int(v) for v in g[f"c{i}_things"]])


def test_oracle_matches_reference_fixtures():
    for i, g, c in _cases():
        mp = P.upsample_and_crop(g[f"c{i}_masks"], c["padded"], c["image"], c["out"])
        torch.testing.assert_close(mp, g[f"c{i}_mask_pred"], atol=1e-5, rtol=1e-5)
        torch.testing.assert_close(P.semantic_inference(g[f"c{i}_cls"], mp), g[f"c{i}_sem"], atol=1e-5, rtol=1e-5)
        seg, info = P.panoptic_inference(g[f"c{i}_cls"], mp, c["C"], c["thr"], c["ovl"], c["things"])
        assert torch.equal(seg, g[f"c{i}_pan"].to(torch.int32))
        assert [[d["id"], int(d["isthing"]), d["category_id"]] for d in info] == g[f"c{i}_info"].tolist()
        assert len(info) >= 3                      # the fixtures exercise real segments (merge, overlap drop, no-object, low score)
        inst = P.instance_inference(g[f"c{i}_cls"], mp, c["C"], int(g[f"c{i}_inst_topk"]), bool(int(g[f"c{i}_inst_panoptic_on"])), c["things"])
        order = torch.argsort(inst["scores"], descending=True)          # topk(sorted=False): compare in score order
        torch.testing.assert_close(inst["scores"][order], g[f"c{i}_inst_scores"], atol=1e-6, rtol=1e-5)
        assert torch.equal(inst["pred_classes"][order], g[f"c{i}_inst_classes"])
        torch.testing.assert_close(inst["pred_masks"][order].flatten(1).sum(1), g[f"c{i}_inst_area"])


def test_oracle_matches_reference_live():
    from oracle import ref_loader
    if not ref_loader.available():
        pytest.skip("reference tree not present")
    O = ref_loader.load_meta_arch().OneFormer
    for seed in (5, 6):
        cls, masks = P.synthetic_predictions(14, 7, 12, 20, seed)
        mp = P.upsample_and_crop(masks, (48, 80), (44, 80), (44, 80))
        fake = types.SimpleNamespace(sem_seg_head=types.SimpleNamespace(num_classes=7), object_mask_threshold=0.4, overlap_threshold=0.7,
                                     metadata=types.SimpleNamespace(thing_dataset_id_to_contiguous_id={5: 5, 6: 6}))
        torch.testing.assert_close(P.semantic_inference(cls, mp), O.semantic_inference(fake, cls, mp), atol=1e-6, rtol=1e-6)
        seg, info = P.panoptic_inference(cls, mp, 7, 0.4, 0.7, [5, 6])
        seg2, info2 = O.panoptic_inference(fake, cls, mp)
        assert torch.equal(seg, seg2) and info == info2

