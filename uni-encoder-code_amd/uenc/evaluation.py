"""The evaluation loop around the hot path: reference model/evaluation/evaluator.py:16-99 (DatasetEvaluator,
DatasetEvaluators), :107-213 (`inference_on_dataset`: the timed `outputs = model(inputs)` loop of §3.1) and :216-229
(inference_context).  Same protocol and log lines (the "Total inference time ... s / iter per device" line is parsed by grep
upstream).  Evaluators: `SemSegEvaluator` (confusion-matrix mIoU / pixel accuracy as detectron2.evaluation.SemSegEvaluator reports
them [not in reference]) and the two depth evaluators of the "sequence" branch, whose arithmetic is plain numpy in the reference:
`KITTIDepthEvaluator` (kitti_evaluation.py:71-279: velodyne ground truth, Eigen crop, median scaling, the seven depth metrics of
`compute_errors` :282-299) and `CityscapesDepthEvaluator` (cityscapes_evaluation.py:231-362).  The reference's Cityscapes instance /
semantic and COCO evaluators wrap third-party scorers (cityscapesscripts, pycocotools, panopticapi) that are not installable here: their
names resolve and raise when constructed.  `inference_on_dataset`, `compute_errors`, the KITTI depth-map projection and the KITTI
evaluation are pinned by fixtures generated from the reference's own functions (oracle/make_data_eval_golden.py ->
tests/golden/data_eval.npz, tests/test_data_eval_cpu.py).
"""
import datetime
import logging
import os
import time
from collections import OrderedDict
from contextlib import ExitStack, contextmanager
from typing import List, Optional, Union

import numpy as np
import torch
from torch import nn


class DatasetEvaluator:
    """evaluator.py:16-63: reset / process(inputs, outputs) / evaluate() -> dict."""

    def reset(self):
        pass

    def process(self, inputs, outputs):
        pass

    def evaluate(self):
        pass


class DatasetEvaluators(DatasetEvaluator):
    """evaluator.py:66-99: dispatches to several evaluators; result keys must not collide."""

    def __init__(self, evaluators):
        super().__init__()
        self._evaluators = evaluators

    def reset(self):
        for e in self._evaluators:
            e.reset()

    def process(self, inputs, outputs):
        for e in self._evaluators:
            e.process(inputs, outputs)

    def evaluate(self):
        results = OrderedDict()
        for e in self._evaluators:
            r = e.evaluate()
            if _is_main_process() and r is not None:
                for k, v in r.items():
                    assert k not in results, "Different evaluators produce results with the same key {}".format(k)
                    results[k] = v
        return results


def _world_size() -> int:
    import torch.distributed as dist
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def _is_main_process() -> bool:
    import torch.distributed as dist
    return (not (dist.is_available() and dist.is_initialized())) or dist.get_rank() == 0


@contextmanager
def inference_context(model):
    """evaluator.py:216-229: eval mode inside, previous mode restored."""
    training_mode = model.training
    model.eval()
    yield
    model.train(training_mode)


def inference_on_dataset(model, data_loader, evaluator: Union[DatasetEvaluator, List[DatasetEvaluator], None], stats: Optional[dict] = None):
    """evaluator.py:107-213.  Runs `model` over `data_loader` under no_grad / eval mode, feeds the evaluator, times data loading,
    compute (synchronised) and evaluation per iteration after min(5, total - 1) warm-up iterations, logs the reference's lines and
    returns `evaluator.evaluate()` ({} when that is None).  `stats`, if given, receives the per-iteration seconds."""
    logger = logging.getLogger(__name__)
    num_devices = _world_size()
    total = len(data_loader)
    logger.info("Start inference on {} batches".format(total))
    if evaluator is None:
        evaluator = DatasetEvaluators([])
    if isinstance(evaluator, (list, tuple)):
        evaluator = DatasetEvaluators(list(evaluator))
    evaluator.reset()
    num_warmup = min(5, total - 1)
    start_time = time.perf_counter()
    total_data_time = total_compute_time = total_eval_time = 0.0
    last_log = 0.0
    with ExitStack() as stack:
        if isinstance(model, nn.Module):
            stack.enter_context(inference_context(model))
        stack.enter_context(torch.no_grad())
        start_data_time = time.perf_counter()
        for idx, inputs in enumerate(data_loader):
            total_data_time += time.perf_counter() - start_data_time
            if idx == num_warmup:
                start_time = time.perf_counter()
                total_data_time = total_compute_time = total_eval_time = 0.0
            start_compute_time = time.perf_counter()
            outputs = model(inputs)
            if torch.cuda.is_available():
                torch.cuda.synchronize()
            total_compute_time += time.perf_counter() - start_compute_time
            start_eval_time = time.perf_counter()
            evaluator.process(inputs, outputs)
            total_eval_time += time.perf_counter() - start_eval_time
            iters_after_start = idx + 1 - num_warmup * int(idx >= num_warmup)
            total_seconds_per_iter = (time.perf_counter() - start_time) / iters_after_start
            if (idx >= num_warmup * 2 or total_compute_time / iters_after_start > 5) and time.perf_counter() - last_log > 5:
                last_log = time.perf_counter()
                eta = datetime.timedelta(seconds=int(total_seconds_per_iter * (total - idx - 1)))
                logger.info(f"Inference done {idx + 1}/{total}. Dataloading: {total_data_time / iters_after_start:.4f} s/iter. "
                            f"Inference: {total_compute_time / iters_after_start:.4f} s/iter. Eval: {total_eval_time / iters_after_start:.4f} s/iter. "
                            f"Total: {total_seconds_per_iter:.4f} s/iter. ETA={eta}")
            start_data_time = time.perf_counter()
    total_time = time.perf_counter() - start_time
    n = max(total - num_warmup, 1)
    # NOTE this format is parsed by grep
    logger.info("Total inference time: {} ({:.6f} s / iter per device, on {} devices)".format(
        str(datetime.timedelta(seconds=total_time)), total_time / n, num_devices))
    logger.info("Total inference pure compute time: {} ({:.6f} s / iter per device, on {} devices)".format(
        str(datetime.timedelta(seconds=int(total_compute_time))), total_compute_time / n, num_devices))
    if stats is not None:
        stats.update(total_s_per_iter=total_time / n, compute_s_per_iter=total_compute_time / n, data_s_per_iter=total_data_time / n,
                     eval_s_per_iter=total_eval_time / n, iterations=n, warmup=num_warmup)
    results = evaluator.evaluate()
    return {} if results is None else results


class SemSegEvaluator(DatasetEvaluator):
    """mIoU / fwIoU / mACC / pACC of the "sem_seg" outputs against "sem_seg" ground-truth maps held by the dataset dicts (key
    `gt_key`, an (H, W) integer array or tensor), accumulated as an (N+1) x (N+1) confusion matrix like
    detectron2.evaluation.SemSegEvaluator; single process (the sharded multi-rank gather is Detectron2 plumbing)."""

    def __init__(self, num_classes: int, ignore_label: int = 255, gt_key: str = "sem_seg_gt"):
        self._num_classes, self._ignore_label, self._gt_key = num_classes, ignore_label, gt_key
        self.reset()

    def reset(self):
        self._conf = np.zeros((self._num_classes + 1, self._num_classes + 1), dtype=np.int64)

    def process(self, inputs, outputs):
        for inp, out in zip(inputs, outputs):
            pred = out["sem_seg"].argmax(dim=0).to("cpu").numpy().astype(np.int64)
            gt = np.asarray(inp[self._gt_key]).astype(np.int64)
            gt[gt == self._ignore_label] = self._num_classes
            self._conf += np.bincount((self._num_classes + 1) * pred.reshape(-1) + gt.reshape(-1),
                                      minlength=self._conf.size).reshape(self._conf.shape)

    def evaluate(self):
        N = self._num_classes
        tp = self._conf.diagonal()[:-1].astype(np.float64)
        pos_gt = np.sum(self._conf[:-1, :-1], axis=0).astype(np.float64)
        pos_pred = np.sum(self._conf[:-1, :-1], axis=1).astype(np.float64)
        class_weights = pos_gt / max(np.sum(pos_gt), 1.0)
        acc_valid = pos_gt > 0
        acc = np.where(acc_valid, tp / np.maximum(pos_gt, 1.0), np.nan)
        union = pos_gt + pos_pred - tp
        iou_valid = acc_valid & (union > 0)
        iou = np.where(iou_valid, tp / np.maximum(union, 1.0), np.nan)
        res = {"mIoU": 100 * float(np.nansum(iou) / max(np.sum(iou_valid), 1)), "fwIoU": 100 * float(np.nansum(iou * class_weights)),
               "mACC": 100 * float(np.nansum(acc) / max(np.sum(acc_valid), 1)), "pACC": 100 * float(np.sum(tp) / max(np.sum(pos_gt), 1.0))}
        return OrderedDict({"sem_seg": res})


# ---------------------------------------------------------------------------------------------------------------------
# depth metrics of the "sequence" branch
# ---------------------------------------------------------------------------------------------------------------------
def compute_errors(gt: np.ndarray, pred: np.ndarray):
    """cityscapes_evaluation.py:365-383 = kitti_evaluation.py:282-299: (abs_rel, sq_rel, rmse, rmse_log, a1, a2, a3) of predicted
    against ground-truth depths (1-D arrays of valid pixels, same dtype arithmetic as numpy gives the inputs)."""
    ratio = np.maximum(gt / pred, pred / gt)
    a1, a2, a3 = ((ratio < 1.25 ** k).mean() for k in (1, 2, 3))
    diff = gt - pred
    rmse = np.sqrt((diff ** 2).mean())
    rmse_log = np.sqrt(((np.log(gt) - np.log(pred)) ** 2).mean())
    return np.mean(np.abs(diff) / gt), np.mean(diff ** 2 / gt), rmse, rmse_log, a1, a2, a3


def disp_to_depth(disp, min_depth=0.1, max_depth=100.0):
    """monodepth_loss.py:103-112: sigmoid disparity -> (scaled disparity, depth)."""
    min_disp, max_disp = 1 / max_depth, 1 / min_depth
    scaled = min_disp + (max_disp - min_disp) * disp
    return scaled, 1 / scaled


def _resize_bilinear(a: np.ndarray, width: int, height: int) -> np.ndarray:
    """cv2.resize(a, (width, height)) with its default INTER_LINEAR [cv2 is not in the reference tree]: bilinear taps at half-pixel
    centres, clamped at the borders, no antialiasing = F.interpolate(..., mode="bilinear", align_corners=False)."""
    t = torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32))[None, None]
    return torch.nn.functional.interpolate(t, size=(height, width), mode="bilinear", align_corners=False)[0, 0].numpy()


_DEPTH_KEYS = ("abs_rel", "sq_rel", "rmse", "rmse_log", "a1", "a2", "a3")


class _DepthEvaluator(DatasetEvaluator):
    """Shared protocol of the two depth evaluators: process() collects (ground-truth depth map, prediction) pairs, evaluate() scores
    every image after median scaling and averages the seven metrics over images -> {"depth_error": {...}} on the main process.
    (The reference parks the pairs as .npy files in a temporary directory shared by the ranks of one machine; here they stay in
    memory and are gathered.)"""
    MIN_DEPTH = 1e-3
    MAX_DEPTH = 80

    def __init__(self, dataset_name=None):
        self._dataset_name = dataset_name
        self._logger = logging.getLogger(__name__)
        self._pairs = []

    def reset(self):
        self._pairs = []

    def _score(self, depth_gt: np.ndarray, pred: np.ndarray):
        raise NotImplementedError

    def evaluate(self):
        import torch.distributed as dist
        pairs = self._pairs
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            box = [None] * dist.get_world_size()
            dist.all_gather_object(box, pairs)
            pairs = [p for part in box for p in part]
        if not _is_main_process():
            return None
        rows = [self._score(gt, pred) for gt, pred in pairs]
        means = np.mean(np.asarray(rows, dtype=np.float64), axis=0) if rows else np.full(7, np.nan)
        return OrderedDict(depth_error={k: v for k, v in zip(_DEPTH_KEYS, means)})

    def _scaled_errors(self, depth_gt: np.ndarray, depth_pred: np.ndarray):
        """per-image median scaling, clamp to [MIN_DEPTH, MAX_DEPTH], the seven metrics (both arrays: valid pixels only)."""
        depth_pred = depth_pred * (np.median(depth_gt) / np.median(depth_pred))
        return compute_errors(depth_gt, np.clip(depth_pred, self.MIN_DEPTH, self.MAX_DEPTH))


class KITTIDepthEvaluator(_DepthEvaluator):
    """kitti_evaluation.py:71-279.  Ground truth = the frame's velodyne scan projected into camera 2 (`generate_depth_map`, depth =
    the points' forward distance); prediction = 1 / (bilinear resize of the scaled disparity to the ground-truth size); scored on
    0.001 < gt < 80 inside the Eigen crop."""

    @staticmethod
    def load_velodyne_points(filename):
        pts = np.fromfile(filename, dtype=np.float32).reshape(-1, 4)
        pts[:, 3] = 1.0                                   # homogeneous (the fourth column is reflectance in the file)
        return pts

    @staticmethod
    def read_calib_file(path):
        """KITTI calibration text: `key: v v v ...` -> float arrays where every token is numeric, strings otherwise."""
        numeric = set("0123456789.e+- ")
        data = {}
        with open(path, "r") as f:
            for line in f:
                key, value = line.split(":", 1)
                value = value.strip()
                data[key] = value
                if numeric.issuperset(value):
                    try:
                        data[key] = np.array([float(v) for v in value.split(" ")])
                    except ValueError:
                        pass
        return data

    @classmethod
    def generate_depth_map(cls, calib_dir, velo_filename, cam=2, vel_depth=False):
        """kitti_evaluation.py:109-166: project the scan with P_rect_0<cam> R_rect_00 [R|T]_velo->cam, round to pixels (minus 1, as
        the KITTI matlab code), keep points inside the S_rect_02 image; where several points hit one pixel the closest wins
        (duplicates are found through the reference's own linear index `row * (W - 1) + col - 1`)."""
        cam2cam = cls.read_calib_file(os.path.join(calib_dir, "calib_cam_to_cam.txt"))
        v2c = cls.read_calib_file(os.path.join(calib_dir, "calib_velo_to_cam.txt"))
        velo2cam = np.vstack((np.hstack((v2c["R"].reshape(3, 3), v2c["T"][..., np.newaxis])), np.array([0, 0, 0, 1.0])))
        im_shape = cam2cam["S_rect_02"][::-1].astype(np.int32)
        R_cam2rect = np.eye(4)
        R_cam2rect[:3, :3] = cam2cam["R_rect_00"].reshape(3, 3)
        P_velo2im = np.dot(np.dot(cam2cam["P_rect_0" + str(cam)].reshape(3, 4), R_cam2rect), velo2cam)
        velo = cls.load_velodyne_points(velo_filename)
        velo = velo[velo[:, 0] >= 0, :]                   # in front of the image plane (approximation)
        pts = np.dot(P_velo2im, velo.T).T
        pts[:, :2] = pts[:, :2] / pts[:, 2][..., np.newaxis]
        if vel_depth:
            pts[:, 2] = velo[:, 0]
        pts[:, 0] = np.round(pts[:, 0]) - 1
        pts[:, 1] = np.round(pts[:, 1]) - 1
        inside = (pts[:, 0] >= 0) & (pts[:, 1] >= 0) & (pts[:, 0] < im_shape[1]) & (pts[:, 1] < im_shape[0])
        pts = pts[inside, :]
        depth = np.zeros((im_shape[:2]))
        rows, cols = pts[:, 1].astype(int), pts[:, 0].astype(int)
        depth[rows, cols] = pts[:, 2]
        lin = pts[:, 1] * (depth.shape[1] - 1) + pts[:, 0] - 1
        uniq, inv, counts = np.unique(lin, return_inverse=True, return_counts=True)
        for u in np.nonzero(counts > 1)[0]:
            hit = np.nonzero(inv == u)[0]
            depth[rows[hit[0]], cols[hit[0]]] = pts[hit, 2].min()
        depth[depth < 0] = 0
        return depth

    def process(self, inputs, outputs):
        for inp, out in zip(inputs, outputs):
            depth_gt = self.generate_depth_map(inp["calib_path"], inp["velo_file"], 2, True)
            disp, _ = disp_to_depth(out["disp_results"])
            disp = np.asarray(disp.squeeze().detach().float().cpu().numpy())
            self._pairs.append((depth_gt, 1 / _resize_bilinear(disp, depth_gt.shape[1], depth_gt.shape[0])))

    def _score(self, depth_gt, depth_pred):
        h, w = depth_gt.shape[:2]
        mask = np.logical_and(depth_gt > self.MIN_DEPTH, depth_gt < self.MAX_DEPTH)
        crop = np.array([0.40810811 * h, 0.99189189 * h, 0.03594771 * w, 0.96405229 * w]).astype(np.int32)      # Eigen crop
        inside = np.zeros(mask.shape, dtype=bool)
        inside[crop[0]:crop[1], crop[2]:crop[3]] = True
        mask &= inside
        return self._scaled_errors(depth_gt[mask], depth_pred[mask])


class CityscapesDepthEvaluator(_DepthEvaluator):
    """cityscapes_evaluation.py:231-362.  Ground truth = the .npy depth map stored beside the image (`/leftImg8bit/test/` ->
    `/gt_depths/`); its lower quarter (ego vehicle) is cut, the scaled disparity is resized to the rest and inverted, and the
    window [256:, 192:1856] is scored on 0.001 < gt < 80."""

    def process(self, inputs, outputs):
        for inp, out in zip(inputs, outputs):
            gt_path = inp["file_name"].replace("/leftImg8bit/test/", "/gt_depths/").replace(".png", ".npy")
            disp, _ = disp_to_depth(out["disp_results"])
            for d in disp.detach().float().cpu()[:, 0].numpy():
                self._pairs.append((np.load(gt_path), d))

    def _score(self, depth_gt, disp_pred):
        h, w = depth_gt.shape[:2]
        h = int(round(h * 0.75))
        depth_gt = depth_gt[:h]
        depth_pred = 1 / _resize_bilinear(np.squeeze(disp_pred), w, h)
        depth_gt, depth_pred = depth_gt[256:, 192:1856], depth_pred[256:, 192:1856]
        mask = np.logical_and(depth_gt > self.MIN_DEPTH, depth_gt < self.MAX_DEPTH)
        return self._scaled_errors(depth_gt[mask], depth_pred[mask])


def print_csv_format(results):
    """detectron2.evaluation.print_csv_format: copy-pastable metric lines."""
    logger = logging.getLogger(__name__)
    for task, res in results.items():
        if isinstance(res, dict):
            important = [(k, v) for k, v in res.items() if "-" not in k]
            logger.info("copypaste: Task: {}".format(task))
            logger.info("copypaste: " + ",".join([k[0] for k in important]))
            logger.info("copypaste: " + ",".join(["{0:.4f}".format(k[1]) for k in important]))
        else:
            logger.info(f"copypaste: {task}={res}")


def _needs(name: str, dep: str):
    class _Unavailable(DatasetEvaluator):
        __doc__ = f"{name} of the reference wraps `{dep}`, which is not installable in this environment (SURVEY.md §8c)."

        def __init__(self, *a, **k):
            raise NotImplementedError(f"{name} needs the third-party package `{dep}` (out of the hot-path scope, SURVEY.md §8f)")
    _Unavailable.__name__ = _Unavailable.__qualname__ = name
    return _Unavailable


COCOEvaluator = _needs("COCOEvaluator", "pycocotools")
InstanceSegEvaluator = _needs("InstanceSegEvaluator", "pycocotools")
CityscapesInstanceEvaluator = _needs("CityscapesInstanceEvaluator", "cityscapesscripts")
CityscapesSemSegEvaluator = _needs("CityscapesSemSegEvaluator", "cityscapesscripts")
