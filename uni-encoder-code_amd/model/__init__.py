"""Import facade with the reference package's name: `from model import add_common_config, ...`

`train_net.py` / `demo/demo.py` of the reference do `from model import (add_common_config,
add_swin_config, add_uni_encoder_config, ...)` and rely on the import side effect of registering
`OneFormer`, `D2SwinTransformer`, `OneFormerHead`, `MSDeformAttnPixelDecoder` and the transformer
decoder (model/__init__.py:1-23, model/modeling/__init__.py:1-10).  Putting
`uni-encoder-code_amd/` on PYTHONPATH gives those drivers this implementation instead.
"""
from uenc.config import *  # noqa: F401,F403
from uenc.config import __all__ as _cfg_all
from uenc import modeling  # noqa: F401  (registers backbone / heads / decoder)
from uenc.oneformer_model import OneFormer  # noqa: F401
from uenc.evaluation import InstanceSegEvaluator  # noqa: F401  (train_net.py:55-56 imports it from `model`)

__all__ = list(_cfg_all) + ["OneFormer", "modeling", "InstanceSegEvaluator"]
