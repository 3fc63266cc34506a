"""Counterpart of the segmentation call of the reference's `DefaultPredictor` (demo/defaults.py:26-160).

`DefaultPredictor(cfg)(image_rgb_uint8_hwc, task)` builds the model through the registries, resizes the
shortest edge to `INPUT.SEG_MIN_SIZE_TEST` (capped by `SEG_MAX_SIZE_TEST`) like the reference's
`ResizeShortestEdge` (:59-61, 157-158) and runs one forward.  With a `previous_frame` it also makes the
reference's second call (:84-97): both frames resized to 192 x 512, `"type": "sequence"`, and returns that
branch's tensors (`disp_results` -> `depth` through `disp_to_depth`, `motion_mask`, `complete_flow`, `cam_T_cam`)
next to the segmentation outputs.  The reference's colour-mapping of those tensors into PIL images (:98-155, with a
camera file read from a hard-coded path) is visualisation and stays out.
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
    def __call__(self, original_image, task="panoptic", previous_frame=None):
        """original_image (and previous_frame): (H, W, 3) uint8 RGB tensor / array."""
        img = torch.as_tensor(original_image)
        height, width = img.shape[:2]
        chw = img.permute(2, 0, 1).float()
        inputs = {"left_image": self._resize(chw), "height": height, "width": width, "task": f"The task is {task}", "type": "segmentation"}
        out = self.model([inputs])[0]
        if previous_frame is not None:
            small = lambda t: F.interpolate(t[None], size=SEQUENCE_SIZE, mode="bilinear", align_corners=False)[0]
            prev = torch.as_tensor(previous_frame).permute(2, 0, 1).float()
            seq = self.model([{"left_image": small(chw), "left_prev_image": small(prev), "height": height, "width": width,
                               "task": f"The task is {task}", "type": "sequence"}])[0]
            if "disp_results" in seq:
                seq["scaled_disp"], seq["depth"] = disp_to_depth(seq["disp_results"])
            out = {**out, **seq}
        return out


SEQUENCE_SIZE = (192, 512)            # demo/defaults.py:84-86


from .evaluation import disp_to_depth  # noqa: E402,F401  (reference model/modeling/monodepth_loss.py:103-112)
