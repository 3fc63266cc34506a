"""`model.data` of the reference (dataset registration, mappers, test loader): the test-time slice lives in uenc.data."""
from uenc.data import DatasetCatalog, MetadataCatalog  # noqa: F401
