"""reference model/data/datasets/register_kitti.py: names resolve to uenc.datasets; the splits are registered on import."""
import os

from uenc.datasets import get_kitti_sequence_files, load_kitti_sequence, register_all_cityscapes_sequence  # noqa: F401

register_all_cityscapes_sequence(os.getenv("DETECTRON2_DATASETS", "datasets"))
