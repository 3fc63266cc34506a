"""`OneFormer` meta-architecture, segmentation branch — counterpart of reference model/oneformer_model.py.

Registered as `OneFormer` in `META_ARCH_REGISTRY`; `forward(batched_inputs: list[dict]) -> list[dict]`
takes the reference's input dicts (`"type": "segmentation"`, `"left_image"` (3,H,W) RGB uint8/float,
`"task"`, optional `"height"`/`"width"`) and returns per-image dicts with `"sem_seg"` (and the raw
`"pred_logits"` / `"pred_masks"` tensors).  `forward_features(batched_inputs)` exposes the hot path
proper (normalise -> backbone -> head, reference :244-253) for training / benchmarking.

Out of scope here: panoptic / instance post-processing and the `"sequence"`
(depth / pose / motion) branch; a `"sequence"` input raises NotImplementedError.
"""
from typing import List, Tuple

import torch
from torch import nn
from torch.nn import functional as F

from . import ops
from .d2 import META_ARCH_REGISTRY, ImageList, build_backbone, build_sem_seg_head, configurable
from .modeling.transformer_decoder.oneformer_transformer_decoder import MLP
from .tokenizer import Tokenize


@META_ARCH_REGISTRY.register()
class OneFormer(nn.Module):
    @configurable
    def __init__(self, *, backbone, sem_seg_head, task_mlp, num_queries: int, object_mask_threshold: float,
                 overlap_threshold: float, size_divisibility: int, sem_seg_postprocess_before_inference: bool,
                 pixel_mean: Tuple[float], pixel_std: Tuple[float], semantic_on: bool, panoptic_on: bool, instance_on: bool,
                 test_topk_per_image: int, task_seq_len: int, max_seq_len: int, is_demo: bool, **unused):
        super().__init__()
        self.backbone, self.sem_seg_head, self.task_mlp = backbone, sem_seg_head, task_mlp
        self.num_queries = num_queries
        self.overlap_threshold, self.object_mask_threshold = overlap_threshold, object_mask_threshold
        if size_divisibility < 0:
            size_divisibility = self.backbone.size_divisibility
        self.size_divisibility = size_divisibility
        self.sem_seg_postprocess_before_inference = sem_seg_postprocess_before_inference
        self.register_buffer("pixel_mean", torch.Tensor(pixel_mean).view(-1, 1, 1), False)
        self.register_buffer("pixel_std", torch.Tensor(pixel_std).view(-1, 1, 1), False)
        self.semantic_on, self.instance_on, self.panoptic_on = semantic_on, instance_on, panoptic_on
        self.test_topk_per_image = test_topk_per_image
        self.task_tokenizer = Tokenize(max_seq_len=task_seq_len)
        self.is_demo = is_demo
        self._task_cache = {}

    @classmethod
    def from_config(cls, cfg):
        backbone = build_backbone(cfg)
        sem_seg_head = build_sem_seg_head(cfg, backbone.output_shape())
        task_mlp = MLP(cfg.INPUT.TASK_SEQ_LEN, cfg.MODEL.ONE_FORMER.HIDDEN_DIM, cfg.MODEL.ONE_FORMER.HIDDEN_DIM, 2)
        t = cfg.MODEL.TEST
        return {
            "backbone": backbone, "sem_seg_head": sem_seg_head, "task_mlp": task_mlp,
            "num_queries": cfg.MODEL.ONE_FORMER.NUM_OBJECT_QUERIES,
            "object_mask_threshold": t.OBJECT_MASK_THRESHOLD, "overlap_threshold": t.OVERLAP_THRESHOLD,
            "size_divisibility": cfg.MODEL.ONE_FORMER.SIZE_DIVISIBILITY,
            "sem_seg_postprocess_before_inference": (t.SEM_SEG_POSTPROCESSING_BEFORE_INFERENCE or t.PANOPTIC_ON or t.INSTANCE_ON),
            "pixel_mean": cfg.MODEL.PIXEL_MEAN, "pixel_std": cfg.MODEL.PIXEL_STD,
            "semantic_on": t.SEMANTIC_ON, "instance_on": t.INSTANCE_ON, "panoptic_on": t.PANOPTIC_ON,
            "test_topk_per_image": cfg.TEST.DETECTIONS_PER_IMAGE,
            "task_seq_len": cfg.INPUT.TASK_SEQ_LEN, "max_seq_len": cfg.INPUT.MAX_SEQ_LEN, "is_demo": cfg.MODEL.IS_DEMO,
        }

    @property
    def device(self):
        return self.pixel_mean.device

    def _task_tokens(self, task: str) -> torch.Tensor:
        """(1, 77) float token ids on the device; cached per prompt (no host-to-device copy in the steady state,
        which also keeps the step capturable into a HIP graph)."""
        key = (task, str(self.device))
        t = self._task_cache.get(key)
        if t is None:
            t = self.task_tokenizer(task).to(self.device).unsqueeze(0).float()
            self._task_cache[key] = t
        return t

    def forward_features(self, batched_inputs: List[dict]):
        """Normalise, pad to a multiple of 32, task embedding, backbone, head.  Returns (outputs, ImageList)."""
        seg = [x for x in batched_inputs if x["type"] == "segmentation"]
        images = [x["left_image"].to(self.device) for x in seg]
        images = [(x - self.pixel_mean) / self.pixel_std for x in images]
        images = ImageList.from_tensors(images, self.size_divisibility)
        tasks = torch.cat([self._task_tokens(x["task"]) for x in seg], dim=0)
        tasks = self.task_mlp(tasks)
        features = self.backbone(images.tensor)
        outputs, _ = self.sem_seg_head(features, None, tasks)
        return outputs, images

    @staticmethod
    def upsample_masks(pred_masks, size):
        """x4 bilinear upsample of the mask logits to the padded input size (reference :255-263), HIP kernel when no
        gradient is needed (inference / the benchmark's forward), autograd-capable ATen path otherwise."""
        if pred_masks.requires_grad and torch.is_grad_enabled() or size[1] % 4:
            return F.interpolate(pred_masks, size=size, mode="bilinear", align_corners=False)
        from . import kernels as K
        return K.upsample_bilinear(pred_masks.detach().float(), size)

    def forward(self, batched_inputs: List[dict]):
        if any(e["type"] == "sequence" for e in batched_inputs):
            raise NotImplementedError("the 'sequence' (depth / pose / motion) branch is out of the hot-path scope, SURVEY.md §8f")
        outputs, images = self.forward_features(batched_inputs)
        mask_cls_results = outputs["pred_logits"]
        mask_pred_results = self.upsample_masks(outputs["pred_masks"], images.tensor.shape[-2:])
        results = []
        seg = [x for x in batched_inputs if x["type"] == "segmentation"]
        for mask_cls, mask_pred, inp, image_size in zip(mask_cls_results, mask_pred_results, seg, images.image_sizes):
            height, width = inp.get("height", image_size[0]), inp.get("width", image_size[1])
            r = {"pred_logits": mask_cls, "pred_masks": mask_pred}
            if self.semantic_on:
                mp = mask_pred[:, : image_size[0], : image_size[1]]
                if (height, width) != tuple(image_size):
                    mp = F.interpolate(mp[None], size=(height, width), mode="bilinear", align_corners=False)[0]
                r["sem_seg"] = self.semantic_inference(mask_cls, mp)
            results.append(r)
        return results

    @staticmethod
    def semantic_inference(mask_cls, mask_pred):
        """reference oneformer_model.py semantic_inference: softmax over classes (drop no-object) x sigmoid masks."""
        mask_cls = F.softmax(mask_cls, dim=-1)[..., :-1]
        return torch.einsum("qc,qhw->chw", mask_cls, mask_pred.sigmoid())
