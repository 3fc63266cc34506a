"""`OneFormer` meta-architecture — counterpart of reference model/oneformer_model.py.

Registered as `OneFormer` in `META_ARCH_REGISTRY`; `forward(batched_inputs: list[dict]) -> list[dict]`
takes the reference's input dicts (`"type": "segmentation"`, `"left_image"` (3,H,W) RGB uint8/float,
`"task"`, optional `"height"`/`"width"`) and returns per-image dicts with `"sem_seg"` (and the raw
`"pred_logits"` / `"pred_masks"` tensors).  `forward_features(batched_inputs)` exposes the hot path
proper (normalise -> backbone -> head, reference :244-253) for training / benchmarking.

Inference post-processing (semantic / panoptic / instance) runs on the device, fused with the mask upsample where no gradient is
needed (csrc/postproc.hip).  `"type": "sequence"` inputs (`"left_image"` + `"left_prev_image"`) take the depth / pose / motion branch
of reference :306-365: two more passes of the accelerated backbone, the ego-pose decoder, the two motion decoders and the TransDSSL
depth decoder (uenc/modeling/{pose_decoder,motion_decoder,pixel_decoder/transdssl}.py; channel counts hard-wired to Swin-T as in the
reference), convolutions on the HIP GEMMs (uenc/convnet.py).
"""
from typing import List, Tuple

import torch
from torch import nn
from torch.nn import functional as F

from . import ops
from .d2 import META_ARCH_REGISTRY, Boxes, ImageList, Instances, build_backbone, build_sem_seg_head, configurable
from .modeling.motion_decoder.dynamo_motion_decoder_mod import MotionDecoderV2
from .modeling.pose_decoder.resnet_like_pose_decoder import ResNetLike
from .modeling.transformer_decoder.oneformer_transformer_decoder import MLP
from .tokenizer import Tokenize


def mask_bounding_boxes(masks: torch.Tensor) -> torch.Tensor:
    """detectron2.structures.BitMasks.get_bounding_boxes [not in reference] for (N, H, W) boolean masks: (x1, y1, x2 + 1, y2 + 1) of the
    pixels that are set, zeros for an empty mask; float32 on the masks' device.  All masks at once (Detectron2 loops over them with
    a torch.where per mask and axis)."""
    n, h, w = masks.shape
    out = torch.zeros((n, 4), dtype=torch.float32, device=masks.device)
    if n == 0:
        return out
    xs, ys = masks.any(dim=1), masks.any(dim=2)                       # (N, W), (N, H)
    has = xs.any(dim=1)
    first = lambda a: a.to(torch.uint8).argmax(dim=1)                 # index of the first True (0 for an all-False row)
    last = lambda a: a.shape[1] - 1 - a.flip(1).to(torch.uint8).argmax(dim=1)
    box = torch.stack([first(xs), first(ys), last(xs) + 1, last(ys) + 1], dim=1).to(torch.float32)
    return torch.where(has[:, None], box, out)


@META_ARCH_REGISTRY.register()
class OneFormer(nn.Module):
    @configurable
    def __init__(self, *, backbone, sem_seg_head, task_mlp, pose_decoder=None, motion_decoder=None, motion_mask=None, depth_on: bool = True,
                 num_queries: int, object_mask_threshold: float,
                 overlap_threshold: float, size_divisibility: int, sem_seg_postprocess_before_inference: bool,
                 pixel_mean: Tuple[float], pixel_std: Tuple[float], semantic_on: bool, panoptic_on: bool, instance_on: bool,
                 test_topk_per_image: int, task_seq_len: int, max_seq_len: int, is_demo: bool, detection_on: bool = False, **unused):
        super().__init__()
        self.backbone, self.sem_seg_head, self.task_mlp = backbone, sem_seg_head, task_mlp
        # the sequence branch's decoders (reference :66-68, built unconditionally at :143-145: they are part of the state dict)
        self.pose_decoder, self.motion_decoder, self.motion_mask = pose_decoder, motion_decoder, motion_mask
        self.depth_on = depth_on
        self.num_queries = num_queries
        self.overlap_threshold, self.object_mask_threshold = overlap_threshold, object_mask_threshold
        if size_divisibility < 0:
            size_divisibility = self.backbone.size_divisibility
        self.size_divisibility = size_divisibility
        self.sem_seg_postprocess_before_inference = sem_seg_postprocess_before_inference
        self.register_buffer("pixel_mean", torch.Tensor(pixel_mean).view(-1, 1, 1), False)
        self.register_buffer("pixel_std", torch.Tensor(pixel_std).view(-1, 1, 1), False)
        self.semantic_on, self.instance_on, self.panoptic_on = semantic_on, instance_on, panoptic_on
        self.detection_on = detection_on        # reference :121, :301-304, :478-480: "box_instances" with boxes taken from the binary masks
        self.test_topk_per_image = test_topk_per_image
        self.task_tokenizer = Tokenize(max_seq_len=task_seq_len)
        self.is_demo = is_demo
        self._task_cache = {}
        # contiguous ids of the 'thing' classes (reference: MetadataCatalog.get(cfg.DATASETS.TRAIN[0]).thing_dataset_id_to_contiguous_id,
        # oneformer_model.py:124, 405); Cityscapes: classes 11..18.  Settable by the driver that knows the dataset.
        self.thing_ids = tuple(range(11, 19))

    @classmethod
    def from_config(cls, cfg):
        backbone = build_backbone(cfg)
        sem_seg_head = build_sem_seg_head(cfg, backbone.output_shape())
        task_mlp = MLP(cfg.INPUT.TASK_SEQ_LEN, cfg.MODEL.ONE_FORMER.HIDDEN_DIM, cfg.MODEL.ONE_FORMER.HIDDEN_DIM, 2)
        t = cfg.MODEL.TEST
        return {
            "backbone": backbone, "sem_seg_head": sem_seg_head, "task_mlp": task_mlp,
            "pose_decoder": ResNetLike(), "motion_decoder": MotionDecoderV2(num_input_images=2, out_dim=3),
            "motion_mask": MotionDecoderV2(num_input_images=2, out_dim=1), "depth_on": t.DEPTH_ON,
            "num_queries": cfg.MODEL.ONE_FORMER.NUM_OBJECT_QUERIES,
            "object_mask_threshold": t.OBJECT_MASK_THRESHOLD, "overlap_threshold": t.OVERLAP_THRESHOLD,
            "size_divisibility": cfg.MODEL.ONE_FORMER.SIZE_DIVISIBILITY,
            "sem_seg_postprocess_before_inference": (t.SEM_SEG_POSTPROCESSING_BEFORE_INFERENCE or t.PANOPTIC_ON or t.INSTANCE_ON),
            "pixel_mean": cfg.MODEL.PIXEL_MEAN, "pixel_std": cfg.MODEL.PIXEL_STD,
            "semantic_on": t.SEMANTIC_ON, "instance_on": t.INSTANCE_ON, "panoptic_on": t.PANOPTIC_ON, "detection_on": t.DETECTION_ON,
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
        results = []
        if any(e["type"] == "segmentation" for e in batched_inputs):
            results = self._forward_segmentation(batched_inputs)
        if any(e["type"] == "sequence" for e in batched_inputs):
            results.append(self._forward_sequence([e for e in batched_inputs if e["type"] == "sequence"]))
        return results

    @torch.no_grad()
    def _forward_sequence(self, seq: List[dict]) -> dict:
        """reference :306-365: depth, ego pose and motion of (previous, current) frame pairs.  One result dict for the whole batch,
        like the reference (`processed_results.append({})` once, :307)."""
        from .modeling.geometry import transformation_from_parameters
        if self.pose_decoder is None or self.sem_seg_head.depth_decoder is None:
            raise NotImplementedError("this model was built without the sequence-branch decoders")
        norm = lambda key: ImageList.from_tensors([(x[key].to(self.device) - self.pixel_mean) / self.pixel_std for x in seq], self.size_divisibility)
        cur, prev = norm("left_image"), norm("left_prev_image")
        f_cur = self.backbone(cur.tensor)
        f_prev = self.backbone(prev.tensor)
        f_motion = {k: torch.cat([f_prev[k].float(), v.float()], dim=1) for k, v in f_cur.items()}
        axisangle, translation = self.pose_decoder(f_motion)
        axisangle, translation = axisangle[:, 0], translation[:, 0]
        cam = transformation_from_parameters(axisangle, translation, invert=True)
        motion = {"motion_input": {"full_res_input": torch.cat([prev.tensor, cur.tensor], dim=1), **f_motion}}
        ego = torch.cat((translation, axisangle), -1).permute(0, 2, 1).unsqueeze(3)
        flow = self.motion_decoder(motion, ego)
        mask = self.motion_mask(motion, ego)
        dummy_tasks = torch.zeros((cur.tensor.shape[0], self.task_mlp.layers[1].out_features), device=self.device)
        _, depth = self.sem_seg_head(None, f_cur, dummy_tasks)
        out = {}
        if self.depth_on:
            out = {"disp_results": depth[("disp", 0)], "motion_mask": mask[("motion_mask", 0)], "complete_flow": flow[("complete_flow", 0)],
                   "cam_T_cam": cam}
        return out

    def _forward_segmentation(self, batched_inputs: List[dict]):
        outputs, images = self.forward_features(batched_inputs)
        mask_cls_results = outputs["pred_logits"]
        padded = tuple(images.tensor.shape[-2:])
        seg = [x for x in batched_inputs if x["type"] == "segmentation"]
        # Inference without gradients: the post-processing kernels interpolate the low-resolution mask logits on the fly, the
        # (Q, H, W) upsampled masks (1.25 GB per 1024 x 2048 image) are never written (csrc/postproc.hip).  With gradients enabled,
        # or when the requested output resolution differs from the image size (a second resize), the reference's sequence of
        # separate passes runs: upsample (:255-263), crop + resize (sem_seg_postprocess), inference.
        fused_ok = not (torch.is_grad_enabled() and outputs["pred_masks"].requires_grad)
        mask_pred_results = None
        results = []
        for i, (mask_cls, inp, image_size) in enumerate(zip(mask_cls_results, seg, images.image_sizes)):
            height, width = inp.get("height", image_size[0]), inp.get("width", image_size[1])
            r = {"pred_logits": mask_cls}
            if fused_ok and (height, width) == tuple(image_size):
                ml = outputs["pred_masks"][i].detach().float().contiguous()
                if self.semantic_on:
                    from . import kernels as K
                    r["sem_seg"] = K.postproc_semantic(ml, F.softmax(mask_cls.detach().float(), dim=-1)[..., :-1], padded, tuple(image_size))
                if self.panoptic_on:
                    r["panoptic_seg"] = self.panoptic_inference_fused(mask_cls.detach().float(), ml, padded, tuple(image_size))
                if self.instance_on:
                    r["instances"] = self.instance_inference(mask_cls.detach().float(), ml, inp["task"], padded, tuple(image_size))
                if self.detection_on:
                    r["box_instances"] = self.instance_inference(mask_cls.detach().float(), ml, inp["task"], padded, tuple(image_size))
            else:
                if mask_pred_results is None:
                    mask_pred_results = self.upsample_masks(outputs["pred_masks"], padded)
                def postprocess(t):      # detectron2 sem_seg_postprocess: crop the padding, resize to the requested resolution
                    t = t[:, : image_size[0], : image_size[1]]
                    if (height, width) != tuple(image_size):
                        t = F.interpolate(t[None], size=(height, width), mode="bilinear", align_corners=False)[0]
                    return t
                mp = postprocess(mask_pred_results[i])
                r["pred_masks"] = mp
                if self.semantic_on:
                    if self.sem_seg_postprocess_before_inference:
                        r["sem_seg"] = self.semantic_inference(mask_cls, mp)
                    else:                # reference :283-289: inference on the padded masks, then crop / resize of the class map
                        r["sem_seg"] = postprocess(self.semantic_inference(mask_cls, mask_pred_results[i]))
                if self.panoptic_on:
                    r["panoptic_seg"] = self.panoptic_inference(mask_cls, mp)
                if self.instance_on:
                    r["instances"] = self.instance_inference(mask_cls, mp, inp["task"])
                if self.detection_on:
                    r["box_instances"] = self.instance_inference(mask_cls, mp, inp["task"])
            results.append(r)
        return results

    @staticmethod
    def semantic_inference(mask_cls, mask_pred):
        """reference oneformer_model.py:367-371: softmax over classes (drop no-object) x sigmoid masks."""
        mask_cls = F.softmax(mask_cls, dim=-1)[..., :-1]
        return torch.einsum("qc,qhw->chw", mask_cls, mask_pred.sigmoid())

    def _segments(self, labels, scores, keep, area, orig, inter):
        """The per-query decisions of panoptic_inference (:399-432) from host copies of the three pixel counts: returns the
        segment id of every query (0 = dropped) and segments_info."""
        segid, info, current, stuff = [0] * len(labels), [], 0, {}
        for q in range(len(labels)):
            if not keep[q]:
                continue
            c = int(labels[q])
            isthing = c in self.thing_ids
            if area[q] > 0 and orig[q] > 0 and inter[q] > 0:
                if area[q] / orig[q] < self.overlap_threshold:
                    continue
                if not isthing:
                    if c in stuff:
                        segid[q] = stuff[c]
                        continue
                    stuff[c] = current + 1
                current += 1
                segid[q] = current
                info.append({"id": current, "isthing": bool(isthing), "category_id": c})
        return segid, info

    def panoptic_inference_fused(self, mask_cls, mask_logits, padded_size, out_size):
        """panoptic_inference (:373-434) on the LOW-resolution mask logits: argmax map and the three per-query pixel counts in one
        kernel, ONE device-to-host copy (the reference: three `.item()` syncs per kept query), then the labelling kernel."""
        from . import kernels as K
        scores, labels = F.softmax(mask_cls, dim=-1).max(-1)
        keep = labels.ne(self.sem_seg_head.num_classes) & (scores > self.object_mask_threshold)
        ids, counts = K.postproc_panoptic_stats(mask_logits, torch.where(keep, scores, torch.zeros_like(scores)), padded_size, out_size)
        host = torch.cat([counts.flatten().float(), labels.float(), keep.float()]).cpu()          # the one synchronisation
        Q = labels.numel()
        area, orig, inter = (host[i * Q:(i + 1) * Q].long().tolist() for i in range(3))
        segid, info = self._segments(host[3 * Q:4 * Q].long().tolist(), None, host[4 * Q:].bool().tolist(), area, orig, inter)
        seg = K.postproc_panoptic_label(mask_logits, ids, torch.tensor(segid, dtype=torch.int32).to(ids.device), padded_size)
        return seg, info

    def instance_inference(self, mask_cls, mask_pred, task_type, padded_size=None, image_size=None):
        """reference oneformer_model.py:436-489 (detection off, not the ADE20K re-indexing).  mask_pred: the (Q, H, W) post-processed
        mask logits, or -- with padded_size / image_size -- the LOW-resolution logits: then only the selected (<= top-k) queries are
        upsampled and cropped (the reference upsamples all Q and gathers).  The class filter of the panoptic setting is one
        torch.isin instead of a Python loop with a sync per detection."""
        C = self.sem_seg_head.num_classes
        scores = F.softmax(mask_cls, dim=-1)[:, :-1]
        labels = torch.arange(C, device=mask_cls.device).unsqueeze(0).repeat(scores.shape[0], 1).flatten(0, 1)
        scores_per_image, topk_indices = scores.flatten(0, 1).topk(self.test_topk_per_image, sorted=False)
        labels_per_image = labels[topk_indices]
        qidx = topk_indices // C
        if self.is_demo:
            keep = scores_per_image > self.object_mask_threshold
            scores_per_image, labels_per_image, qidx = scores_per_image[keep], labels_per_image[keep], qidx[keep]
        if self.panoptic_on:
            keep = torch.isin(labels_per_image, torch.tensor(self.thing_ids, device=labels_per_image.device))
            scores_per_image, labels_per_image, qidx = scores_per_image[keep], labels_per_image[keep], qidx[keep]
        if padded_size is not None:
            from . import kernels as K
            sel = mask_pred[qidx].contiguous()
            if sel.shape[0] > 0 and padded_size[1] % 4 == 0:
                mask_pred = K.upsample_bilinear(sel[None], padded_size)[0][:, : image_size[0], : image_size[1]]
            else:
                mask_pred = F.interpolate(sel[None], size=tuple(padded_size), mode="bilinear", align_corners=False)[0][:, : image_size[0], : image_size[1]]
        else:
            mask_pred = mask_pred[qidx]
        result = Instances(tuple(mask_pred.shape[-2:]))
        result.pred_masks = (mask_pred > 0).float()
        # reference :478-482: with MODEL.TEST.DETECTION_ON the boxes are the tight boxes of the binary masks, else zeros
        result.pred_boxes = Boxes(mask_bounding_boxes(mask_pred > 0) if getattr(self, "detection_on", False) else torch.zeros(mask_pred.size(0), 4))
        flat = result.pred_masks.flatten(1)
        mask_scores_per_image = (mask_pred.sigmoid().flatten(1) * flat).sum(1) / (flat.sum(1) + 1e-6)
        result.scores = scores_per_image * mask_scores_per_image
        result.pred_classes = labels_per_image
        return result

    def panoptic_inference(self, mask_cls, mask_pred):
        """reference oneformer_model.py:373-434 on materialised (Q, H, W) mask logits (the fallback path)."""
        scores, labels = F.softmax(mask_cls, dim=-1).max(-1)
        prob = mask_pred.sigmoid()
        keep = labels.ne(self.sem_seg_head.num_classes) & (scores > self.object_mask_threshold)
        h, w = prob.shape[-2:]
        seg = torch.zeros((h, w), dtype=torch.int32, device=prob.device)
        if int(keep.sum()) == 0:
            return seg, []
        ids = (torch.where(keep, scores, torch.zeros_like(scores)).view(-1, 1, 1) * prob).argmax(0)
        over = prob >= 0.5
        Q = labels.numel()
        onehot = ids.flatten()
        area = torch.bincount(onehot, minlength=Q)
        orig = over.flatten(1).sum(1)
        won = over.flatten(1).gather(0, onehot[None])[0]
        inter = torch.bincount(onehot, weights=won.float(), minlength=Q).long()
        segid, info = self._segments(labels.tolist(), None, keep.tolist(), area.tolist(), orig.tolist(), inter.tolist())
        sid = torch.tensor(segid, dtype=torch.int32, device=prob.device)
        seg = torch.where(won.view(h, w), sid[ids], torch.zeros_like(seg))
        return seg, info
