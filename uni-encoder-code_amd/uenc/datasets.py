"""Dataset registration of the hot path's caller: file lists -> Detectron2 dataset dicts in the DatasetCatalog.

Counterparts of reference model/data/datasets/register_cityscapes_panoptic.py (:22-51 file scan, :54-115 dict building, :118-141 split
table, :144-206 metadata) and model/data/datasets/register_kitti.py (:22-70 file scan, :73-92 dicts, :95-110 registration, :113-126
splits).  `"segmentation"` dicts feed DatasetMapper.process_segmentation_data, `"sequence"` dicts process_sequence_data (uenc/data.py).

`CITYSCAPES_CATEGORIES` is detectron2.data.datasets.builtin_meta's table [not in reference]: the 19 evaluation classes of
Cityscapes with their label ids, train ids and palette colours (the public cityscapesscripts label definition).

Registration is lazy (the catalog stores a function), so registering under a root that does not exist is harmless -- as in the
reference, which registers under $DETECTRON2_DATASETS (default "datasets") at import time.
"""
import json
import logging
import os
from typing import Dict, List, Optional, Tuple

from .data import DatasetCatalog, MetadataCatalog

logger = logging.getLogger(__name__)

# (name, label id, train id, is thing, colour)
_CS = [("road", 7, 0, 0, (128, 64, 128)), ("sidewalk", 8, 1, 0, (244, 35, 232)), ("building", 11, 2, 0, (70, 70, 70)),
       ("wall", 12, 3, 0, (102, 102, 156)), ("fence", 13, 4, 0, (190, 153, 153)), ("pole", 17, 5, 0, (153, 153, 153)),
       ("traffic light", 19, 6, 0, (250, 170, 30)), ("traffic sign", 20, 7, 0, (220, 220, 0)), ("vegetation", 21, 8, 0, (107, 142, 35)),
       ("terrain", 22, 9, 0, (152, 251, 152)), ("sky", 23, 10, 0, (70, 130, 180)), ("person", 24, 11, 1, (220, 20, 60)),
       ("rider", 25, 12, 1, (255, 0, 0)), ("car", 26, 13, 1, (0, 0, 142)), ("truck", 27, 14, 1, (0, 0, 70)), ("bus", 28, 15, 1, (0, 60, 100)),
       ("train", 31, 16, 1, (0, 80, 100)), ("motorcycle", 32, 17, 1, (0, 0, 230)), ("bicycle", 33, 18, 1, (119, 11, 32))]
CITYSCAPES_CATEGORIES = [{"color": c, "isthing": t, "id": i, "trainId": ti, "name": n} for n, i, ti, t, c in _CS]

_IMG_SUFFIX = "_leftImg8bit.png"


def _catalog_names():
    return DatasetCatalog.list() if hasattr(DatasetCatalog, "list") else list(DatasetCatalog.keys())


def _register(name, loader, **metadata):
    if name in _catalog_names():
        DatasetCatalog.remove(name)
    DatasetCatalog.register(name, loader)
    MetadataCatalog.get(name).set(**metadata)


# ---------------------------------------------------------------------------------------------------------------------
# Cityscapes panoptic ("segmentation" dicts)
# ---------------------------------------------------------------------------------------------------------------------
def get_cityscapes_panoptic_files(image_dir: str, gt_dir: str, json_info: dict) -> List[Tuple[str, str, list]]:
    """(image file, panoptic label file, segments_info) per annotation of the panoptic json, images found by scanning
    image_dir/<city>/<id>_leftImg8bit.png (register_cityscapes_panoptic.py:22-51)."""
    cities = sorted(os.listdir(image_dir))
    logger.info(f"{len(cities)} cities found in '{image_dir}'.")
    by_id: Dict[str, str] = {}
    for city in cities:
        for basename in sorted(os.listdir(os.path.join(image_dir, city))):
            assert basename.endswith(_IMG_SUFFIX), basename
            by_id[basename[: -len(_IMG_SUFFIX)]] = os.path.join(image_dir, city, basename)
    files = []
    for ann in json_info["annotations"]:
        image_file = by_id.get(ann["image_id"])
        assert image_file is not None, "No image {} found for annotation {}".format(ann["image_id"], ann["file_name"])
        files.append((image_file, os.path.join(gt_dir, ann["file_name"]), ann["segments_info"]))
    assert len(files), "No images found in {}".format(image_dir)
    assert os.path.isfile(files[0][0]), files[0][0]
    assert os.path.isfile(files[0][1]), files[0][1]
    return files


def load_cityscapes_panoptic(image_dir: str, gt_dir: str, gt_json: str, meta: dict) -> List[dict]:
    """Dataset dicts {"file_name", "type": "segmentation", "image_id", "sem_seg_file_name", "pan_seg_file_name", "segments_info"} with
    category ids mapped to contiguous train ids (register_cityscapes_panoptic.py:54-115)."""
    assert os.path.exists(gt_json), \
        "Please run `python cityscapesscripts/preparation/createPanopticImgs.py` to generate label files."
    with open(gt_json) as f:
        json_info = json.load(f)
    thing_map, stuff_map = meta["thing_dataset_id_to_contiguous_id"], meta["stuff_dataset_id_to_contiguous_id"]

    def contiguous(seg: dict) -> dict:
        cid = seg["category_id"]
        seg["category_id"] = thing_map[cid] if cid in thing_map else stuff_map[cid]
        return seg

    ret = []
    for image_file, label_file, segments_info in get_cityscapes_panoptic_files(image_dir, gt_dir, json_info):
        stem = os.path.splitext(os.path.basename(image_file))[0]
        ret.append({"file_name": image_file, "type": "segmentation", "image_id": "_".join(stem.split("_")[:3]),
                    "sem_seg_file_name": image_file.replace("leftImg8bit", "gtFine").split(".")[0] + "_labelTrainIds.png",
                    "pan_seg_file_name": label_file, "segments_info": [contiguous(s) for s in segments_info]})
    assert len(ret), f"No images found in {image_dir}!"
    assert os.path.isfile(ret[0]["sem_seg_file_name"]), \
        "Please generate labelTrainIds.png with cityscapesscripts/preparation/createTrainIdLabelImgs.py"
    assert os.path.isfile(ret[0]["pan_seg_file_name"]), \
        "Please generate panoptic annotation with python cityscapesscripts/preparation/createPanopticImgs.py"
    return ret


def _panoptic_splits() -> Dict[str, Tuple[str, str, str]]:
    out = {}
    for prefix, top in (("cityscapes_fine_panoptic", "cityscapes"), ("cityscapes_segmentation_crop_fine_panoptic", "cityscapes_crop")):
        for split in ("train", "val"):
            out[f"{prefix}_{split}"] = (f"{top}/leftImg8bit/{split}", f"{top}/gtFine/cityscapes_panoptic_{split}",
                                        f"{top}/gtFine/cityscapes_panoptic_{split}.json")
    return out


_RAW_CITYSCAPES_PANOPTIC_SPLITS = _panoptic_splits()


def cityscapes_panoptic_meta() -> dict:
    """thing_* and stuff_* both list all 19 classes (Detectron2's visualiser convention); the two id maps split label id -> train id
    by `isthing` (register_cityscapes_panoptic.py:144-188)."""
    names = [k["name"] for k in CITYSCAPES_CATEGORIES]
    colors = [k["color"] for k in CITYSCAPES_CATEGORIES]
    return {"thing_classes": names, "thing_colors": colors, "stuff_classes": list(names), "stuff_colors": list(colors),
            "thing_dataset_id_to_contiguous_id": {k["id"]: k["trainId"] for k in CITYSCAPES_CATEGORIES if k["isthing"] == 1},
            "stuff_dataset_id_to_contiguous_id": {k["id"]: k["trainId"] for k in CITYSCAPES_CATEGORIES if k["isthing"] != 1}}


def register_all_cityscapes_panoptic(root: str):
    meta = cityscapes_panoptic_meta()
    for key, (image_dir, gt_dir, gt_json) in _RAW_CITYSCAPES_PANOPTIC_SPLITS.items():
        image_dir, gt_dir, gt_json = (os.path.join(root, p) for p in (image_dir, gt_dir, gt_json))
        _register(key, lambda x=image_dir, y=gt_dir, z=gt_json: load_cityscapes_panoptic(x, y, z, meta),
                  panoptic_root=gt_dir, image_root=image_dir, panoptic_json=gt_json, gt_dir=gt_dir.replace("cityscapes_panoptic_", ""),
                  evaluator_type="cityscapes_panoptic_seg", ignore_label=255, label_divisor=1000, **meta)


# ---------------------------------------------------------------------------------------------------------------------
# KITTI raw sequences ("sequence" dicts)
# ---------------------------------------------------------------------------------------------------------------------
_SIDE = {"2": 2, "3": 3, "l": 2, "r": 3}


def get_kitti_sequence_files(data_root: str, files_list: str, img_ext: str = ".jpg") -> List[Tuple[str, Optional[str], Optional[str], str, str, str]]:
    """One (frame, previous frame | None, next frame | None, calibration dir, velodyne file, side) per line "<folder> <frame> <side>" of the
    split file whose frame exists; neighbours are kept only when BOTH exist; a frame without calibration + velodyne data is an error
    (register_kitti.py:22-70)."""
    with open(files_list, "r") as f:
        lines = f.read().splitlines()
    files = []
    for line in lines:
        info = line.split()
        folder = info[0]
        frame, side = (int(info[1]), info[2]) if len(info) == 3 else (0, None)
        cam_dir = os.path.join(data_root, folder, "image_0{}/data".format(_SIDE[side]))
        cur, prev, nxt = (os.path.join(cam_dir, "{:010d}{}".format(frame + d, img_ext)) for d in (0, -1, 1))
        calib_path = os.path.join(data_root, folder.split("/")[0])
        velo = os.path.join(data_root, folder, "velodyne_points/data/{:010d}.bin".format(frame))
        if not os.path.isfile(cur):
            continue
        if not (os.path.isdir(calib_path) and os.path.isfile(velo)):
            raise NotImplementedError
        both = os.path.isfile(prev) and os.path.isfile(nxt)
        files.append((cur, prev if both else None, nxt if both else None, calib_path, velo, side))
    assert len(files), "No images found in {}".format(data_root)
    return files


def load_kitti_sequence(data_root: str, files_list: str, img_ext: str = ".jpg") -> List[dict]:
    ret = [{"type": "sequence", "file_name": cur, "image_id": os.path.splitext(os.path.basename(cur))[0], "left_prev_image_file": prev,
            "left_nxt_image_file": nxt, "calib_path": calib, "velo_file": velo, "side": side}
           for cur, prev, nxt, calib, velo, side in get_kitti_sequence_files(data_root, files_list, img_ext)]
    assert len(ret), f"No images found in {data_root}!"
    return ret


_RAW_CITYSCAPES_SEQUENCE_SPLITS = {          # (the reference's name for the KITTI split table, register_kitti.py:113-124)
    "KITTI_eigen_zhou_train_split": ("kitti_data", "kitti_data/eigen_zhou_train_files_kitti.txt", ".jpg"),
    "KITTI_standard_eigen_test_split": ("kitti_data", "kitti_data/standard_eigen_test_files.txt", ".jpg"),
}


def register_all_cityscapes_sequence(root: str):
    for key, (data_root, files_list, ext) in _RAW_CITYSCAPES_SEQUENCE_SPLITS.items():
        data_root, files_list = os.path.join(root, data_root), os.path.join(root, files_list)
        _register(key, lambda x=data_root, y=files_list, z=ext: load_kitti_sequence(x, y, z), left_image_root=data_root, evaluator_type="kitti_depth")


def register_all(root: Optional[str] = None):
    """What importing `model.data.datasets` does in the reference: both tables under $DETECTRON2_DATASETS (default "datasets")."""
    root = os.getenv("DETECTRON2_DATASETS", "datasets") if root is None else root
    register_all_cityscapes_panoptic(root)
    register_all_cityscapes_sequence(root)
