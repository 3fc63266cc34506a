"""`from model.data.build import *` (train_net.py:64): reference model/data/build.py exports build_detection_test_loader."""
from uenc.data import build_detection_test_loader  # noqa: F401

__all__ = ["build_detection_test_loader"]
