"""`build_pixel_decoder` — same dispatch and error as reference model/modeling/pixel_decoder/fpn.py:23-35."""
from ...d2 import SEM_SEG_HEADS_REGISTRY


def build_pixel_decoder(cfg, input_shape, depth_decoder=False):
    name = cfg.MODEL.SEM_SEG_HEAD.PIXEL_DECODER_NAME if not depth_decoder else cfg.MODEL.SEM_SEG_HEAD.DEPTH_DECODER_NAME
    model = SEM_SEG_HEADS_REGISTRY.get(name)(cfg, input_shape)
    forward_features = getattr(model, "forward_features", None)
    if not callable(forward_features):
        raise ValueError(
            "Only SEM_SEG_HEADS with forward_features method can be used as pixel decoder. "
            f"Please implement forward_features for {name} to only return mask features.")
    return model
