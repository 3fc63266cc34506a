"""OneFormerHead — counterpart of reference model/modeling/meta_arch/oneformer_head.py:21-149.

Same registry name, `from_config` keys and `forward(seg_features, depth_features, tasks, mask=None)
-> (predictions_seg, predictions_depth)` contract.  The depth decoder (`TransDSSL`, sequence branch)
 is built, like in the reference (:112), from `MODEL.SEM_SEG_HEAD.DEPTH_DECODER_NAME` whenever a class of
that name is registered (uenc/modeling/pixel_decoder/transdssl.py registers `TransDSSL`); otherwise `depth_decoder` is None and
depth features are rejected loudly.
"""
import logging
from typing import Dict

from torch import nn

from ...d2 import SEM_SEG_HEADS_REGISTRY, ShapeSpec, configurable
from ..pixel_decoder.fpn import build_pixel_decoder
from ..transformer_decoder.oneformer_transformer_decoder import build_transformer_decoder


@SEM_SEG_HEADS_REGISTRY.register()
class OneFormerHead(nn.Module):
    _version = 2

    def _load_from_state_dict(self, state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys, error_msgs):
        version = local_metadata.get("version", None)
        if version is None or version < 2:
            for k in list(state_dict.keys()):          # pre-v2 checkpoints kept pixel-decoder weights at the head's top level
                newk = k
                if "sem_seg_head" in k and not k.startswith(prefix + "predictor") and not k.startswith("sem_seg_head.depth_decoder."):
                    newk = k.replace(prefix, prefix + "pixel_decoder.").replace("pixel_decoder.pixel_decoder.", "pixel_decoder.")
                if newk != k:
                    state_dict[newk] = state_dict.pop(k)
                    logging.getLogger(__name__).warning("converted legacy key %s", k)
        super()._load_from_state_dict(state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys, error_msgs)

    @configurable
    def __init__(self, input_shape: Dict[str, ShapeSpec], *, num_classes: int, pixel_decoder: nn.Module,
                 depth_decoder=None, loss_weight: float = 1.0, ignore_value: int = -1,
                 transformer_predictor: nn.Module, transformer_in_feature: str):
        super().__init__()
        input_shape = sorted(input_shape.items(), key=lambda x: x[1].stride)
        self.in_features = [k for k, v in input_shape]
        self.ignore_value, self.common_stride, self.loss_weight = ignore_value, 4, loss_weight
        self.pixel_decoder = pixel_decoder
        self.depth_decoder = depth_decoder
        self.predictor = transformer_predictor
        self.transformer_in_feature = transformer_in_feature
        self.num_classes = num_classes

    @classmethod
    def from_config(cls, cfg, input_shape: Dict[str, ShapeSpec]):
        tif = cfg.MODEL.ONE_FORMER.TRANSFORMER_IN_FEATURE
        if tif in ("transformer_encoder", "multi_scale_pixel_decoder"):
            in_ch = cfg.MODEL.SEM_SEG_HEAD.CONVS_DIM
        elif tif == "pixel_embedding":
            in_ch = cfg.MODEL.SEM_SEG_HEAD.MASK_DIM
        else:
            in_ch = input_shape[tif].channels
        depth_name = cfg.MODEL.SEM_SEG_HEAD.DEPTH_DECODER_NAME
        depth_decoder = build_pixel_decoder(cfg, input_shape, depth_decoder=True) if depth_name in SEM_SEG_HEADS_REGISTRY else None
        return {
            "input_shape": {k: v for k, v in input_shape.items() if k in cfg.MODEL.SEM_SEG_HEAD.IN_FEATURES},
            "ignore_value": cfg.MODEL.SEM_SEG_HEAD.IGNORE_VALUE,
            "num_classes": cfg.MODEL.SEM_SEG_HEAD.NUM_CLASSES,
            "pixel_decoder": build_pixel_decoder(cfg, input_shape, depth_decoder=False),
            "depth_decoder": depth_decoder,
            "loss_weight": cfg.MODEL.SEM_SEG_HEAD.LOSS_WEIGHT,
            "transformer_in_feature": tif,
            "transformer_predictor": build_transformer_decoder(cfg, in_ch, mask_classification=True),
        }

    def forward(self, seg_features, depth_features, tasks, mask=None):
        return self.layers(seg_features, depth_features, tasks, mask)

    def layers(self, seg_features, depth_features, tasks, mask=None):
        predictions_seg, predictions_depth = {}, {}
        if seg_features is not None and tasks is not None:
            mask_features, transformer_encoder_features, multi_scale_features = self.pixel_decoder.forward_features(seg_features)
            if self.transformer_in_feature == "multi_scale_pixel_decoder":
                predictions_seg = self.predictor(multi_scale_features, mask_features, tasks, mask)
            else:
                raise NotImplementedError(
                    f"TRANSFORMER_IN_FEATURE={self.transformer_in_feature!r}: only 'multi_scale_pixel_decoder' "
                    "(every shipped config, oneformer_R50_bs16_90k.yaml:21) is on the hot path")
        if depth_features is not None:
            if self.depth_decoder is None:
                raise NotImplementedError("depth decoder (sequence branch) is out of the hot-path scope, SURVEY.md §8f")
            predictions_depth = self.depth_decoder.forward_features(depth_features)
        return predictions_seg, predictions_depth
