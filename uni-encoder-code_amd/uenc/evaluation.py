"""The evaluation loop around the hot path: reference model/evaluation/evaluator.py:16-99 (DatasetEvaluator,
DatasetEvaluators), :107-213 (`inference_on_dataset`: the timed `outputs = model(inputs)` loop of §3.1) and :216-229
(inference_context).  Same protocol and log lines (the "Total inference time ... s / iter per device" line is parsed by grep
upstream).  One self-contained evaluator is provided, `SemSegEvaluator` (confusion-matrix mIoU / pixel accuracy as
detectron2.evaluation.SemSegEvaluator reports them [not in reference]); the reference's Cityscapes / COCO / KITTI evaluators wrap
third-party scorers (cityscapesscripts, pycocotools, panopticapi) that are not installable here: their names resolve and raise
when constructed.
"""
import datetime
import logging
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
CityscapesDepthEvaluator = _needs("CityscapesDepthEvaluator", "cityscapesscripts")
KITTIDepthEvaluator = _needs("KITTIDepthEvaluator", "the KITTI depth ground truth tooling")
