"""reference model/data/datasets/register_cityscapes_panoptic.py: names resolve to uenc.datasets; the splits are registered on import."""
import os

from uenc.datasets import (CITYSCAPES_CATEGORIES, get_cityscapes_panoptic_files, load_cityscapes_panoptic,  # noqa: F401
                           register_all_cityscapes_panoptic)

register_all_cityscapes_panoptic(os.getenv("DETECTRON2_DATASETS", "datasets"))
