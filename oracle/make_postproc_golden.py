"""tests/golden/postproc.npz from the reference's own post-processing methods (build container only; python -m oracle.make_postproc_golden).

TEST INFRASTRUCTURE.  Calls `OneFormer.semantic_inference` / `OneFormer.panoptic_inference` of /root/reference/model/
oneformer_model.py (:367-434) unbound on a namespace `self`, on the synthetic predictions of oracle/postproc_ref.py, after the
reference's own upsample (:258-263) and detectron2's documented sem_seg_postprocess; stores inputs + outputs only."""
import os
import types

import numpy as np
import torch
import torch.nn.functional as F

from . import postproc_ref as P, ref_loader

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "postproc.npz")
CASES = [  # (Q, C, low-res h, w, padded size, image size, output size, object threshold, overlap threshold, thing ids, seed)
    (12, 5, 16, 24, (64, 96), (64, 96), (64, 96), 0.5, 0.8, (3, 4), 0),
    (20, 19, 24, 32, (96, 128), (90, 120), (90, 120), 0.8, 0.8, tuple(range(11, 19)), 1),
    (9, 4, 8, 12, (32, 48), (30, 41), (45, 60), 0.3, 0.6, (2, 3), 2),
]


def main():
    O = ref_loader.load_meta_arch().OneFormer
    post = __import__("sys").modules["detectron2.modeling.postprocessing"].sem_seg_postprocess
    arrs = {"ncases": np.int64(len(CASES))}
    for i, (Q, C, h, w, padded, image, out, thr, ovl, things, seed) in enumerate(CASES):
        cls, masks = P.synthetic_predictions(Q, C, h, w, seed)
        fake = types.SimpleNamespace(sem_seg_head=types.SimpleNamespace(num_classes=C), object_mask_threshold=thr, overlap_threshold=ovl,
                                     metadata=types.SimpleNamespace(thing_dataset_id_to_contiguous_id={t: t for t in things}))
        up = F.interpolate(masks[None], size=padded, mode="bilinear", align_corners=False)[0]            # :258-263
        mp = post(up, image, out[0], out[1])                                                              # :277-279 (before inference)
        sem = O.semantic_inference(fake, cls, mp)
        seg, info = O.panoptic_inference(fake, cls, mp)
        fake.num_queries, fake.test_topk_per_image, fake.is_demo, fake.panoptic_on, fake.detection_on = Q, min(10, Q * C), False, i != 1, False
        fake.device, fake.metadata.name = torch.device("cpu"), "cityscapes_fine_panoptic_val"
        inst = O.instance_inference(fake, cls, mp, "The task is instance")                                 # :436-489
        order = torch.argsort(inst.scores, descending=True)
        arrs.update({f"c{i}_cls": cls.numpy(), f"c{i}_masks": masks.numpy(),
                     f"c{i}_meta": np.array([Q, C, h, w, *padded, *image, *out], dtype=np.int64), f"c{i}_thr": np.array([thr, ovl]),
                     f"c{i}_things": np.array(things, dtype=np.int64), f"c{i}_mask_pred": mp.numpy(), f"c{i}_sem": sem.numpy(),
                     f"c{i}_inst_scores": inst.scores[order].numpy(), f"c{i}_inst_classes": inst.pred_classes[order].numpy(),
                     f"c{i}_inst_area": inst.pred_masks[order].flatten(1).sum(1).numpy(), f"c{i}_inst_topk": np.int64(fake.test_topk_per_image),
                     f"c{i}_inst_panoptic_on": np.int64(int(fake.panoptic_on)),
                     f"c{i}_pan": seg.numpy(), f"c{i}_info": np.array([[d["id"], int(d["isthing"]), d["category_id"]] for d in info],
                                                                     dtype=np.int64).reshape(-1, 3)})
        print(f"case {i}: {len(info)} segments, ids {sorted(set(seg.flatten().tolist()))}")
    np.savez_compressed(OUT, **arrs)
    print("wrote", OUT, os.path.getsize(OUT), "bytes")


if __name__ == "__main__":
    main()
