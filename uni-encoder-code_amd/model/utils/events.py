"""`from model.utils.events import MLflowWriter, set_environment_variables, setup_mlflow` (train_net.py:63).  Experiment tracking is
outside the hot path ("plain JSON lines from the bench harness"); the names resolve so the driver imports, and say what
is missing when used."""


def _no_mlflow(*a, **k):
    raise NotImplementedError("MLflow logging (reference model/utils/events.py:179-254) needs the `mlflow` package: out of the hot-path scope")


class MLflowWriter:
    def __init__(self, *a, **k):
        _no_mlflow()


set_environment_variables = setup_mlflow = _no_mlflow
