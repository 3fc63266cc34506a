"""Counterpart of the segmentation call of the reference's `DefaultPredictor` (demo/defaults.py:26-160).

`DefaultPredictor(cfg)(image_rgb_uint8_hwc, task)` builds the model through the registries, resizes the
shortest edge to `INPUT.SEG_MIN_SIZE_TEST` (capped by `SEG_MAX_SIZE_TEST`) like the reference's
`ResizeShortestEdge` (:59-61, 157-158) and runs one forward.  The reference's second, 192x512
"sequence" call (:96-97) belongs to the out-of-scope depth branch.
"""
import torch
import torch.nn.functional as F

from .d2 import build_model


class DefaultPredictor:
    def __init__(self, cfg):
        self.cfg = cfg.clone()
        self.model = build_model(self.cfg)
        self.model.eval()
        if getattr(cfg.MODEL, "WEIGHTS", ""):               # demo/defaults.py:56-57
            from .checkpoint import DetectionCheckpointer
            DetectionCheckpointer(self.model).load(cfg.MODEL.WEIGHTS)
        self.min_size, self.max_size = cfg.INPUT.SEG_MIN_SIZE_TEST, cfg.INPUT.SEG_MAX_SIZE_TEST

    def _resize(self, img):  # (3, H, W) float
        _, h, w = img.shape
        scale = self.min_size / min(h, w)
        if max(h, w) * scale > self.max_size:
            scale = self.max_size / max(h, w)
        nh, nw = int(h * scale + 0.5), int(w * scale + 0.5)
        if (nh, nw) == (h, w):
            return img
        return F.interpolate(img[None], size=(nh, nw), mode="bilinear", align_corners=False)[0]

    @torch.no_grad()
    def __call__(self, original_image, task="panoptic"):
        """original_image: (H, W, 3) uint8 RGB tensor / array."""
        img = torch.as_tensor(original_image)
        height, width = img.shape[:2]
        img = self._resize(img.permute(2, 0, 1).float())
        inputs = {"left_image": img, "height": height, "width": width, "task": f"The task is {task}", "type": "segmentation"}
        return self.model([inputs])[0]
