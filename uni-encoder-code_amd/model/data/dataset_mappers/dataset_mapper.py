"""`from model.data.dataset_mappers.dataset_mapper import DatasetMapper` (train_net.py:65)."""
from uenc.data import DatasetMapper, build_augmentation, read_image  # noqa: F401
