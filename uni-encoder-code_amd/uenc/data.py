"""Input pipeline of the hot path's caller: dataset dicts -> model inputs, and the test loader.

Counterpart of reference model/data/dataset_mappers/dataset_mapper.py:25-50 (build_augmentation), :81-224 (DatasetMapper), :244-289
(process_segmentation_data) and model/data/build.py:27-120 (build_detection_test_loader), i.e. what `Trainer.build_test_loader`
(train_net.py:176-185) and `inference_on_dataset` consume.  The Detectron2 pieces they rest on (DatasetCatalog, MetadataCatalog,
ResizeShortestEdge, read_image, InferenceSampler, DatasetFromList / MapDataset, trivial_batch_collator) are [not in reference]
and are restated here in the slice the test-time path uses.

Scope: the test-time mappings -- `"segmentation"` (what the evaluation loop feeds OneFormer.forward) and `"sequence"` (:290-332:
current / previous / next frame for the depth / pose / motion branch, OneFormer._forward_sequence).  Training-time augmentation and
annotation transforms belong to the reference's training drivers, which cannot run as shipped: they raise
NotImplementedError here.  Dataset registration (Cityscapes panoptic, KITTI sequences): uenc/datasets.py.
"""
import copy
import logging
from typing import Any, Callable, Dict, List, Optional, Sequence, Union

import numpy as np
import torch
import torch.utils.data as torchdata

logger = logging.getLogger(__name__)

try:  # pragma: no cover - not installable in the build image
    import detectron2
    if not (hasattr(detectron2, "__version__") and getattr(detectron2, "__file__", None) is not None):      # a test's name holder
        raise ImportError("detectron2 is not installed")
    from detectron2.data import DatasetCatalog, MetadataCatalog
    HAVE_D2 = True
except Exception:
    HAVE_D2 = False

if not HAVE_D2:
    class _Metadata:
        def __init__(self, name):
            object.__setattr__(self, "name", name)

        def set(self, **kw):
            for k, v in kw.items():
                setattr(self, k, v)
            return self

        def get(self, key, default=None):
            return getattr(self, key, default)

        def as_dict(self):
            return dict(self.__dict__)

    class _MetadataCatalog(dict):
        """name -> metadata object, created on first access (detectron2.data.MetadataCatalog semantics)."""

        def get(self, name):
            if name not in self:
                self[name] = _Metadata(name)
            return self[name]

    class _DatasetCatalog(dict):
        """name -> function returning list[dict] (detectron2.data.DatasetCatalog semantics)."""

        def register(self, name, func):
            assert callable(func), "You must register a function with `DatasetCatalog.register`!"
            assert name not in self, f"Dataset '{name}' is already registered!"
            self[name] = func

        def get(self, name):
            try:
                f = self[name]
            except KeyError as e:
                raise KeyError(f"Dataset '{name}' is not registered! Available datasets are: {', '.join(self.keys())}") from e
            return f()

        def remove(self, name):
            self.pop(name)

    MetadataCatalog = _MetadataCatalog()
    DatasetCatalog = _DatasetCatalog()


# ---------------------------------------------------------------------------------------------------------------------
# image reading + the test-time transform
# ---------------------------------------------------------------------------------------------------------------------
def read_image(file_name: str, format: Optional[str] = None) -> np.ndarray:
    """detectron2.data.detection_utils.read_image: HWC uint8 in `format` ("RGB" / "BGR" / "L"), EXIF orientation applied."""
    from PIL import Image, ImageOps
    with open(file_name, "rb") as f:
        image = Image.open(f)
        image = ImageOps.exif_transpose(image)
        conv = format
        if format == "BGR":
            conv = "RGB"
        if conv is not None:
            image = image.convert(conv)
        arr = np.asarray(image)
    if format == "L":
        arr = np.expand_dims(arr, -1)
    elif format == "BGR":
        arr = arr[:, :, ::-1]
    return arr


_SEQUENCE_SIZE = {"cs": (192, 512), "kitti": (192, 640)}         # (h, w) every frame of a sequence is brought to before augmentation


def read_sequence_image(file_name: str, format: Optional[str] = None, dataset: str = "cs") -> np.ndarray:
    """The mapper's own `read_image` for sequence frames (dataset_mapper.py:53-78): the file is resized to the dataset's fixed
    network size with PIL LANCZOS FIRST (Cityscapes 192 x 512, KITTI 192 x 640), then EXIF-oriented and converted to `format`."""
    from PIL import Image, ImageOps
    if dataset not in _SEQUENCE_SIZE:
        raise NotImplementedError
    h, w = _SEQUENCE_SIZE[dataset]
    with open(file_name, "rb") as f:
        image = Image.open(f).resize((w, h), Image.LANCZOS)
        image = ImageOps.exif_transpose(image)
        conv = "RGB" if format == "BGR" else format
        if conv is not None:
            image = image.convert(conv)
        arr = np.asarray(image)
    if format == "L":
        arr = np.expand_dims(arr, -1)
    elif format == "BGR":
        arr = arr[:, :, ::-1]
    return arr


def check_image_size(dataset_dict: dict, image: np.ndarray):
    """detection_utils.check_image_size: the file must have the size the dataset dict states (and the dict gets it if absent)."""
    if "width" in dataset_dict or "height" in dataset_dict:
        if (image.shape[1], image.shape[0]) != (dataset_dict["width"], dataset_dict["height"]):
            raise ValueError("Mismatched image shape{}, got {}, expect {}.".format(
                " for image " + dataset_dict["file_name"] if "file_name" in dataset_dict else "",
                (image.shape[1], image.shape[0]), (dataset_dict["width"], dataset_dict["height"])))
    dataset_dict.setdefault("width", image.shape[1])
    dataset_dict.setdefault("height", image.shape[0])


class ResizeShortestEdge:
    """detectron2.data.transforms.ResizeShortestEdge with sample_style "choice" (test time: one size): scale the shorter edge to
    `short_edge_length`, cap the longer at `max_size`, round half up, resize with PIL bilinear (as ResizeTransform.apply_image)."""

    def __init__(self, short_edge_length, max_size, sample_style="choice"):
        if isinstance(short_edge_length, int):
            short_edge_length = (short_edge_length, short_edge_length)
        self.short_edge_length, self.max_size, self.sample_style = tuple(short_edge_length), max_size, sample_style

    @staticmethod
    def get_output_shape(oldh: int, oldw: int, short_edge_length: int, max_size: int):
        h, w = oldh, oldw
        scale = short_edge_length * 1.0 / min(h, w)
        newh, neww = (short_edge_length, scale * w) if h < w else (scale * h, short_edge_length)
        if max(newh, neww) > max_size:
            scale = max_size * 1.0 / max(newh, neww)
            newh, neww = newh * scale, neww * scale
        return int(newh + 0.5), int(neww + 0.5)

    def __call__(self, image: np.ndarray) -> np.ndarray:
        from PIL import Image
        if self.sample_style == "range":
            size = int(np.random.randint(self.short_edge_length[0], self.short_edge_length[1] + 1))
        else:
            size = int(np.random.choice(self.short_edge_length))
        if size == 0:
            return image
        newh, neww = self.get_output_shape(image.shape[0], image.shape[1], size, self.max_size)
        if (newh, neww) == image.shape[:2]:
            return image
        assert image.dtype == np.uint8
        squeeze = image.ndim == 3 and image.shape[2] == 1
        pil = Image.fromarray(image[:, :, 0] if squeeze else image)
        out = np.asarray(pil.resize((neww, newh), Image.BILINEAR))
        return np.expand_dims(out, -1) if squeeze else out


def build_augmentation(cfg, is_train: bool, for_segmentation: bool = True) -> list:
    """dataset_mapper.py:25-50, test-time branch: [ResizeShortestEdge(<prefix>MIN_SIZE_TEST, <prefix>MAX_SIZE_TEST, "choice")]."""
    if is_train:
        raise NotImplementedError("training-time augmentation is outside the hot-path scope (SURVEY.md §8f)")
    prefix = "SEG_" if for_segmentation else "DEPTH_"
    return [ResizeShortestEdge(getattr(cfg.INPUT, f"{prefix}MIN_SIZE_TEST"), getattr(cfg.INPUT, f"{prefix}MAX_SIZE_TEST"), "choice")]


class DatasetMapper:
    """dataset_mapper.py:81-289 at test time: a Detectron2 dataset dict {"file_name", "type", ...} -> the dict OneFormer.forward
    takes: "left_image" (3, H, W) uint8 tensor after the test-time resize, "task" = "The task is {panoptic|semantic|instance}",
    "height" / "width" = the ORIGINAL size (the resolution the outputs are returned at), annotations dropped."""

    def __init__(self, cfg=None, is_train: bool = False, *, seg_augmentations=None, dep_augmentations=None, image_format: str = "RGB",
                 task: str = "panoptic"):
        if is_train:
            raise NotImplementedError("the training-time mapper is outside the hot-path scope (SURVEY.md §8f)")
        if cfg is not None:                                   # from_config (dataset_mapper.py:181-222)
            seg_augmentations = build_augmentation(cfg, False, for_segmentation=True)
            dep_augmentations = build_augmentation(cfg, False, for_segmentation=False)
            image_format = cfg.INPUT.FORMAT
            task = cfg.MODEL.TEST.TASK
        assert task in ["panoptic", "semantic", "instance"]
        self.is_train, self.seg_augmentations, self.image_format, self.task = False, list(seg_augmentations or []), image_format, task
        self.dep_augmentations = list(dep_augmentations or [])
        logger.info("[DatasetMapper] Augmentations used in inference for segmentations: %s", self.seg_augmentations)
        logger.info("[DatasetMapper] Augmentations used in inference for depth: %s", self.dep_augmentations)

    def __call__(self, dataset_dict: dict) -> dict:
        dataset_dict = copy.deepcopy(dataset_dict)
        kind = dataset_dict.get("type")
        if kind == "segmentation":
            return self.process_segmentation_data(dataset_dict)
        if kind == "sequence":
            return self.process_sequence_data(dataset_dict)
        raise ValueError("Unknown dataset type: {}".format(kind))

    def process_sequence_data(self, dataset_dict: dict) -> dict:
        """dataset_mapper.py:290-332: the current frame and -- when the dict names them -- its previous and next frames, each read at the
        KITTI network size (the reference hard-codes dataset="kitti": 192 x 640) and passed through the SAME depth test transform;
        -> "left_image" / "left_prev_image" / "left_next_image" (3, H, W) uint8.  "height" / "width" are checked against / set from
        the resized frame, as the reference's check_image_size calls do.  (The reference's `getattr(dataset_dict, "cam_info_file")`
        is a dict-attribute lookup that can never succeed, so no "baseline" key is ever produced; none is produced here.)"""
        frames = {"left_image": read_sequence_image(dataset_dict["file_name"], format=self.image_format, dataset="kitti")}
        check_image_size(dataset_dict, frames["left_image"])
        if dataset_dict["left_prev_image_file"] is not None:
            for key, src in (("left_prev_image", "left_prev_image_file"), ("left_next_image", "left_nxt_image_file")):
                frames[key] = read_sequence_image(dataset_dict[src], format=self.image_format, dataset="kitti")
                check_image_size(dataset_dict, frames[key])
        for key, image in frames.items():
            for aug in self.dep_augmentations:          # deterministic at test time: every frame gets the same resize
                image = aug(image)
            dataset_dict[key] = torch.as_tensor(np.ascontiguousarray(image.transpose(2, 0, 1)))
        return dataset_dict

    def process_segmentation_data(self, dataset_dict: dict) -> dict:
        image = read_image(dataset_dict["file_name"], format=self.image_format)
        check_image_size(dataset_dict, image)
        dataset_dict["task"] = f"The task is {self.task}"
        for aug in self.seg_augmentations:
            image = aug(image)
        dataset_dict["left_image"] = torch.as_tensor(np.ascontiguousarray(image.transpose(2, 0, 1)))
        dataset_dict.pop("annotations", None)
        dataset_dict.pop("left_sem_seg_file_name", None)
        return dataset_dict


# ---------------------------------------------------------------------------------------------------------------------
# test loader
# ---------------------------------------------------------------------------------------------------------------------
class InferenceSampler(torchdata.Sampler):
    """detectron2.data.samplers.InferenceSampler: every rank gets one contiguous shard; together they cover each index exactly once."""

    def __init__(self, size: int, rank: Optional[int] = None, world_size: Optional[int] = None):
        import torch.distributed as dist
        if rank is None:
            rank = dist.get_rank() if dist.is_available() and dist.is_initialized() else 0
        if world_size is None:
            world_size = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
        assert size > 0
        shard, left = size // world_size, size % world_size
        sizes = [shard + int(r < left) for r in range(world_size)]
        begin = sum(sizes[:rank])
        self._local_indices = range(begin, min(begin + sizes[rank], size))

    def __iter__(self):
        yield from self._local_indices

    def __len__(self):
        return len(self._local_indices)


class _MapDataset(torchdata.Dataset):
    def __init__(self, dataset, map_func):
        self._dataset, self._map_func = dataset, map_func

    def __len__(self):
        return len(self._dataset)

    def __getitem__(self, idx):
        return self._map_func(self._dataset[idx])


def trivial_batch_collator(batch):
    """No collation: the model takes list[dict]."""
    return batch


def get_detection_dataset_dicts(names: Union[str, Sequence[str]], filter_empty: bool = False, proposal_files=None) -> List[dict]:
    if isinstance(names, str):
        names = [names]
    assert len(names), names
    out: List[dict] = []
    for n in names:
        dicts = DatasetCatalog.get(n)
        assert len(dicts), "Dataset '{}' is empty!".format(n)
        out.extend(dicts)
    return out


def build_detection_test_loader(dataset, dataset_name=None, *, mapper: Optional[Callable[[Dict[str, Any]], Any]] = None, sampler=None,
                                batch_size: int = 1, num_workers: int = 0, collate_fn=None) -> torchdata.DataLoader:
    """model/data/build.py:59-120.  Either `(dataset: list[dict] | Dataset, mapper=...)` or, as Trainer.build_test_loader calls it
    (train_net.py:185), `(cfg, dataset_name, mapper=None)`: dataset dicts from the DatasetCatalog, DatasetMapper(cfg, False),
    cfg.DATALOADER.NUM_WORKERS workers.  Batch size 1 per worker, InferenceSampler, no collation."""
    if hasattr(dataset, "DATALOADER") and dataset_name is not None:      # a cfg
        cfg = dataset
        dataset = get_detection_dataset_dicts(dataset_name, filter_empty=False)
        if mapper is None:
            mapper = DatasetMapper(cfg, False)
        num_workers = cfg.DATALOADER.NUM_WORKERS
    if mapper is not None:
        dataset = _MapDataset(dataset, mapper)
    if sampler is None:
        sampler = InferenceSampler(len(dataset))
    return torchdata.DataLoader(dataset, batch_size=batch_size, sampler=sampler, drop_last=False, num_workers=num_workers,
                                collate_fn=trivial_batch_collator if collate_fn is None else collate_fn)
