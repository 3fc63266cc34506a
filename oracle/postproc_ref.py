"""CPU restatement (fp32, plain PyTorch) of the reference's segmentation post-processing.

TEST INFRASTRUCTURE: only `tests/` imports it.  Parity status: PINNED -- `oracle/make_postproc_golden.py` calls the reference's
own `OneFormer.semantic_inference` / `panoptic_inference` / `instance_inference` (model/oneformer_model.py:367-489, loaded through
`oracle/ref_loader.load_meta_arch`) and the documented `sem_seg_postprocess` on synthetic predictions and commits inputs +
outputs as `tests/golden/postproc.npz`; `tests/test_postproc_cpu.py` checks this file against them (and live against the
reference where /root/reference exists).  Citations are relative to /root/reference/model/oneformer_model.py.
"""
from typing import Dict, List, Sequence, Tuple

import torch
import torch.nn.functional as F

Tensor = torch.Tensor


def upsample_and_crop(mask_logits: Tensor, padded_size: Tuple[int, int], image_size: Tuple[int, int], out_size: Tuple[int, int]) -> Tensor:
    """:258-263 bilinear upsample of (Q, h, w) logits to the padded input size, then detectron2's sem_seg_postprocess (:277-279):
    crop the padding, resize to the requested output resolution."""
    m = F.interpolate(mask_logits[None], size=tuple(padded_size), mode="bilinear", align_corners=False)[0]
    m = m[:, : image_size[0], : image_size[1]]
    if tuple(out_size) != tuple(image_size):
        m = F.interpolate(m[None], size=tuple(out_size), mode="bilinear", align_corners=False)[0]
    return m


def semantic_inference(mask_cls: Tensor, mask_pred: Tensor) -> Tensor:
    """:367-371 (Q, C+1) class logits, (Q, H, W) mask logits -> (C, H, W)."""
    p = F.softmax(mask_cls, dim=-1)[..., :-1]
    return torch.einsum("qc,qhw->chw", p, mask_pred.sigmoid())


def panoptic_inference(mask_cls: Tensor, mask_pred: Tensor, num_classes: int, object_mask_threshold: float, overlap_threshold: float,
                       thing_ids: Sequence[int]):
    """:373-434 -> (panoptic_seg (H, W) int32, segments_info)."""
    scores, labels = F.softmax(mask_cls, dim=-1).max(-1)
    prob = mask_pred.sigmoid()
    keep = labels.ne(num_classes) & (scores > object_mask_threshold)
    cur_scores, cur_classes, cur_masks = scores[keep], labels[keep], prob[keep]
    h, w = cur_masks.shape[-2:]
    seg = torch.zeros((h, w), dtype=torch.int32)
    info: List[dict] = []
    if cur_masks.shape[0] == 0:
        return seg, info
    ids = (cur_scores.view(-1, 1, 1) * cur_masks).argmax(0)
    current, stuff = 0, {}
    for k in range(cur_classes.shape[0]):
        c = int(cur_classes[k])
        isthing = c in thing_ids
        mine = ids == k
        area, orig = int(mine.sum()), int((cur_masks[k] >= 0.5).sum())
        m = mine & (cur_masks[k] >= 0.5)
        if area > 0 and orig > 0 and int(m.sum()) > 0:
            if area / orig < overlap_threshold:
                continue
            if not isthing:
                if c in stuff:
                    seg[m] = stuff[c]
                    continue
                stuff[c] = current + 1
            current += 1
            seg[m] = current
            info.append({"id": current, "isthing": bool(isthing), "category_id": c})
    return seg, info


def instance_inference(mask_cls: Tensor, mask_pred: Tensor, num_classes: int, topk: int, panoptic_on: bool, thing_ids: Sequence[int]):
    """:436-489 (is_demo False, detection off, not ADE20K): -> dict(pred_masks (n, H, W) float {0, 1}, scores (n), pred_classes (n)),
    in the order torch.topk(sorted=False) happened to return -- callers compare as sets."""
    Q = mask_cls.shape[0]
    scores = F.softmax(mask_cls, dim=-1)[:, :-1]
    labels = torch.arange(num_classes).unsqueeze(0).repeat(Q, 1).flatten(0, 1)
    s, idx = scores.flatten(0, 1).topk(topk, sorted=False)
    lab = labels[idx]
    mp = mask_pred[idx // num_classes]
    if panoptic_on:
        keep = torch.tensor([int(v) in thing_ids for v in lab], dtype=torch.bool)
        s, lab, mp = s[keep], lab[keep], mp[keep]
    pm = (mp > 0).float()
    ms = (mp.sigmoid().flatten(1) * pm.flatten(1)).sum(1) / (pm.flatten(1).sum(1) + 1e-6)
    return {"pred_masks": pm, "scores": s * ms, "pred_classes": lab}


def synthetic_predictions(Q: int, C: int, h: int, w: int, seed: int = 0):
    """Predictions that exercise every branch of panoptic_inference: coherent blobs (segments that pass the overlap test),
    two 'stuff' queries of one class (merge), a heavily overlapped query (dropped by the overlap threshold), a no-object query
    and a low-score query."""
    g = torch.Generator().manual_seed(seed)
    ys, xs = torch.meshgrid(torch.arange(h, dtype=torch.float32), torch.arange(w, dtype=torch.float32), indexing="ij")
    cls = torch.randn(Q, C + 1, generator=g)
    masks = torch.empty(Q, h, w)
    for q in range(Q):
        cy, cx = float(torch.rand(1, generator=g)) * h, float(torch.rand(1, generator=g)) * w
        ry, rx = h * (0.12 + 0.2 * float(torch.rand(1, generator=g))), w * (0.12 + 0.2 * float(torch.rand(1, generator=g)))
        d = ((ys - cy) / ry) ** 2 + ((xs - cx) / rx) ** 2
        masks[q] = 6.0 * (1.0 - d) + 0.7 * torch.randn(h, w, generator=g)
        cls[q, q % C] += 6.0                                   # confident class
    if Q >= 6:
        cls[1, :] = cls[0, :]                                  # same (stuff) class as query 0: merged when both survive
        masks[2] = masks[0] - 1.0                              # lives inside query 0's blob: loses the argmax -> overlap test
        cls[3, C] += 12.0                                      # no-object
        cls[4, :] = 0.1 * torch.randn(C + 1, generator=g)      # low score
    return cls, masks
