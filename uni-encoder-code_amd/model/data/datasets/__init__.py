"""`model.data.datasets` of the reference: importing it registers the Cityscapes panoptic and KITTI sequence splits (uenc.datasets)."""
from . import register_cityscapes_panoptic, register_kitti  # noqa: F401
