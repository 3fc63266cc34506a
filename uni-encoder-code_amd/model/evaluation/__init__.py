"""`from model.evaluation import ...` (train_net.py:43-48) resolves here: the evaluation loop and evaluator names of uenc.evaluation."""
from uenc.evaluation import (COCOEvaluator, CityscapesDepthEvaluator, CityscapesInstanceEvaluator, DatasetEvaluator,  # noqa: F401
                             DatasetEvaluators, InstanceSegEvaluator, KITTIDepthEvaluator, SemSegEvaluator, inference_context,
                             inference_on_dataset, print_csv_format)
