"""Input pipeline + evaluation loop on the CPU: test-time DatasetMapper, ResizeShortestEdge, InferenceSampler,
build_detection_test_loader, inference_on_dataset, SemSegEvaluator, and the `model.*` import surface train_net.py uses."""
import logging
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _cfg():
    import model  # noqa: F401
    from uenc.config import add_common_config, add_swin_config, add_uni_encoder_config
    from uenc.d2 import get_cfg
    cfg = get_cfg()
    add_common_config(cfg); add_swin_config(cfg); add_uni_encoder_config(cfg)
    return cfg


def _write_images(tmp_path, sizes):
    from PIL import Image
    g = np.random.default_rng(0)
    dicts = []
    for i, (h, w) in enumerate(sizes):
        arr = g.integers(0, 256, (h, w, 3), dtype=np.uint8)
        fn = str(tmp_path / f"img{i}.png")
        Image.fromarray(arr).save(fn)
        dicts.append({"file_name": fn, "height": h, "width": w, "image_id": i, "type": "segmentation",
                      "annotations": [{"bbox": [0, 0, 1, 1]}], "sem_seg_gt": (arr[:, :, 0] % 19).astype(np.int64)})
    return dicts


def test_resize_shortest_edge_sizes():
    from uenc.data import ResizeShortestEdge
    f = ResizeShortestEdge.get_output_shape
    assert f(512, 1024, 384, 1024) == (384, 768)           # the shipped Cityscapes test size (unified_encoder_cityscapes.yaml: 384 / 1024)
    assert f(1024, 2048, 384, 1024) == (384, 768)
    assert f(100, 1000, 384, 1024) == (102, 1024)          # capped by the long edge, rounded half up
    assert f(1000, 100, 384, 1024) == (1024, 102)
    assert f(333, 500, 384, 10000) == (384, 577)
    img = np.zeros((50, 100, 3), dtype=np.uint8); img[:, 50:] = 200
    out = ResizeShortestEdge(25, 1000)(img)
    assert out.shape == (25, 50, 3) and out[:, :20].max() == 0 and out[:, 30:].min() == 200


def test_dataset_mapper_test_time(tmp_path):
    from uenc.data import DatasetMapper
    cfg = _cfg()
    cfg.merge_from_list(["INPUT.SEG_MIN_SIZE_TEST", 32, "INPUT.SEG_MAX_SIZE_TEST", 80, "INPUT.FORMAT", "RGB", "MODEL.TEST.TASK", "semantic"])
    d = _write_images(tmp_path, [(48, 96)])[0]
    out = DatasetMapper(cfg, False)(d)
    assert out["task"] == "The task is semantic" and out["type"] == "segmentation"
    assert out["left_image"].dtype == torch.uint8 and tuple(out["left_image"].shape) == (3, 32, 64)
    assert out["height"] == 48 and out["width"] == 96 and "annotations" not in out            # outputs are asked for at the original size
    assert "annotations" in d                                                              # the input dict was not modified
    # RGB order: channel 0 of the tensor is the file's red channel (checked on an un-resized read)
    from PIL import Image
    raw = np.asarray(Image.open(d["file_name"]))
    same = DatasetMapper(is_train=False, seg_augmentations=[], image_format="RGB", task="panoptic")(d)
    assert np.array_equal(same["left_image"].numpy(), raw.transpose(2, 0, 1))
    bgr = DatasetMapper(is_train=False, seg_augmentations=[], image_format="BGR", task="panoptic")(d)
    assert np.array_equal(bgr["left_image"].numpy()[0], raw[:, :, 2])
    with pytest.raises(ValueError):
        DatasetMapper(cfg, False)({"file_name": d["file_name"], "type": "detection"})
    with pytest.raises(ValueError):                                                        # size check of detection_utils.check_image_size
        DatasetMapper(cfg, False)({**d, "width": 97})
    with pytest.raises(NotImplementedError):
        DatasetMapper(cfg, True)


def test_test_loader_and_sampler(tmp_path):
    from uenc.data import DatasetCatalog, InferenceSampler, build_detection_test_loader
    cfg = _cfg()
    cfg.merge_from_list(["INPUT.SEG_MIN_SIZE_TEST", 32, "INPUT.SEG_MAX_SIZE_TEST", 64, "DATALOADER.NUM_WORKERS", 0])
    dicts = _write_images(tmp_path, [(40, 56)] * 5)
    name = "uenc_test_synthetic_%d" % os.getpid()
    DatasetCatalog.register(name, lambda: dicts)
    loader = build_detection_test_loader(cfg, name)
    batches = list(loader)
    assert len(batches) == 5 and all(isinstance(b, list) and len(b) == 1 for b in batches)
    assert [b[0]["image_id"] for b in batches] == [0, 1, 2, 3, 4]
    # shards over ranks cover every index exactly once
    parts = [list(InferenceSampler(11, rank=r, world_size=4)) for r in range(4)]
    assert sorted(sum(parts, [])) == list(range(11)) and [len(p) for p in parts] == [3, 3, 3, 2]
    with pytest.raises(KeyError):
        DatasetCatalog.get("not_registered")


def test_inference_on_dataset_and_semseg_evaluator(tmp_path, caplog):
    from uenc.data import DatasetMapper, build_detection_test_loader
    from uenc.evaluation import DatasetEvaluators, SemSegEvaluator, inference_on_dataset
    dicts = _write_images(tmp_path, [(24, 32)] * 8)
    mapper = DatasetMapper(is_train=False, seg_augmentations=[], image_format="RGB", task="semantic")
    loader = build_detection_test_loader(dicts, mapper=mapper)

    class Oracle(torch.nn.Module):          # predicts the ground truth for even images, class 0 everywhere for odd ones
        def forward(self, inputs):
            assert not self.training and not torch.is_grad_enabled()
            outs = []
            for x in inputs:
                gt = torch.as_tensor(x["sem_seg_gt"])
                sem = torch.zeros(19, *gt.shape)
                if x["image_id"] % 2 == 0:
                    sem.scatter_(0, gt[None], 1.0)
                else:
                    sem[0] = 1.0
                outs.append({"sem_seg": sem})
            return outs
    m = Oracle().train()
    stats = {}
    with caplog.at_level(logging.INFO, logger="uenc.evaluation"):
        res = inference_on_dataset(m, loader, [SemSegEvaluator(19)], stats=stats)
    assert m.training                                              # mode restored
    assert any("Total inference time:" in r.message and "s / iter per device, on 1 devices" in r.message for r in caplog.records)
    assert stats["iterations"] == 3 and stats["warmup"] == 5       # min(5, total - 1) warm-up iterations are not timed
    r = res["sem_seg"]
    gt = np.stack([d["sem_seg_gt"] for d in dicts])
    want_pacc = 100.0 * (np.sum(gt[0::2] == gt[0::2]) + np.sum(gt[1::2] == 0)) / gt.size
    assert abs(r["pACC"] - want_pacc) < 1e-9 and 0 < r["mIoU"] < 100
    assert inference_on_dataset(m, loader, None) == {}
    with pytest.raises(AssertionError):
        DatasetEvaluators([SemSegEvaluator(19), SemSegEvaluator(19)]).reset() or inference_on_dataset(m, loader, [SemSegEvaluator(19), SemSegEvaluator(19)])


def test_model_package_import_surface():
    """The `model.*` names train_net.py:43-65 and demo/demo.py:26-31 import resolve against the facade."""
    import model
    from model import InstanceSegEvaluator, add_common_config, add_swin_config, add_uni_encoder_config  # noqa: F401
    from model.data.build import build_detection_test_loader  # noqa: F401
    from model.data.dataset_mappers.dataset_mapper import DatasetMapper  # noqa: F401
    from model.evaluation import COCOEvaluator, CityscapesDepthEvaluator, CityscapesInstanceEvaluator, KITTIDepthEvaluator  # noqa: F401
    from model.utils.events import MLflowWriter, set_environment_variables, setup_mlflow  # noqa: F401
    assert hasattr(model, "add_dinat_config")
    for cls in (InstanceSegEvaluator, COCOEvaluator, CityscapesInstanceEvaluator, MLflowWriter):
        with pytest.raises(NotImplementedError):
            cls()
