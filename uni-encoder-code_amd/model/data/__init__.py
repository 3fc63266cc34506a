"""`model.data` of the reference (dataset registration, mappers, test loader): the test-time slice lives in uenc.data."""
from uenc.data import DatasetCatalog, MetadataCatalog  # noqa: F401
from . import datasets  # noqa: E402,F401  (reference model/data/__init__.py:1: importing the package registers the dataset splits)
