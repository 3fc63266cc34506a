"""The slice of the Detectron2 API the hot path is built against.

The reference is a Detectron2 plug-in: its classes register into D2's registries and
are constructed by `build_model(cfg)` from a yacs CfgNode.  When `detectron2` is importable the real
registries / base classes are used, so the classes here drop into an existing D2 process; otherwise
this module provides minimal stand-ins with the same names and semantics (Registry, CfgNode with
`_BASE_` YAML chains, `configurable`, ShapeSpec, Backbone, ImageList, Conv2d, get_norm,
build_backbone / build_sem_seg_head / build_model).
"""
import ast
import copy
import functools
import inspect
import os
from typing import Any, Dict, List, Optional, Tuple

import torch
import torch.nn as nn
import torch.nn.functional as F

try:  # pragma: no cover - not installable in the build image
    import detectron2  # noqa: F401
    HAVE_D2 = hasattr(detectron2, "__version__") and getattr(detectron2, "__file__", None) is not None      # (not a test's name holder)
except Exception:
    HAVE_D2 = False

if HAVE_D2:  # pragma: no cover
    from detectron2.config import CfgNode, configurable, get_cfg
    from detectron2.layers import Conv2d, ShapeSpec, get_norm
    from detectron2.modeling import (BACKBONE_REGISTRY, META_ARCH_REGISTRY, SEM_SEG_HEADS_REGISTRY, Backbone,
                                     build_backbone, build_model, build_sem_seg_head)
    from detectron2.structures import Boxes, ImageList, Instances
    from detectron2.utils.registry import Registry
else:
    class Registry:
        """name -> object map with decorator registration (fvcore.common.registry.Registry semantics)."""

        def __init__(self, name: str):
            self._name = name
            self._obj_map: Dict[str, Any] = {}

        def _do_register(self, name, obj):
            assert name not in self._obj_map, f"An object named '{name}' was already registered in '{self._name}' registry!"
            self._obj_map[name] = obj

        def register(self, obj=None):
            if obj is None:
                def deco(func_or_class):
                    self._do_register(func_or_class.__name__, func_or_class)
                    return func_or_class
                return deco
            self._do_register(obj.__name__, obj)
            return obj

        def get(self, name):
            ret = self._obj_map.get(name)
            if ret is None:
                raise KeyError(f"No object named '{name}' found in '{self._name}' registry!")
            return ret

        def __contains__(self, name):
            return name in self._obj_map

    class CfgNode(dict):
        """Attribute-style nested config with YAML `_BASE_` inheritance and KEY VALUE overrides (yacs subset)."""

        def __init__(self, init=None):
            super().__init__()
            object.__setattr__(self, "_frozen", False)
            for k, v in (init or {}).items():
                self[k] = CfgNode(v) if isinstance(v, dict) and not isinstance(v, CfgNode) else v

        def __getattr__(self, name):
            try:
                return self[name]
            except KeyError:
                raise AttributeError(name)

        def __setattr__(self, name, value):
            if object.__getattribute__(self, "_frozen"):
                raise AttributeError(f"Attempted to set {name} on a frozen CfgNode")
            self[name] = value

        def _set_frozen(self, flag):
            object.__setattr__(self, "_frozen", flag)
            for v in self.values():
                if isinstance(v, CfgNode):
                    v._set_frozen(flag)

        def freeze(self):
            self._set_frozen(True)

        def defrost(self):
            self._set_frozen(False)

        def is_frozen(self):
            return object.__getattribute__(self, "_frozen")

        def clone(self):
            return copy.deepcopy(self)

        def __deepcopy__(self, memo):
            out = CfgNode()
            for k, v in self.items():
                dict.__setitem__(out, k, copy.deepcopy(v, memo))
            return out

        def merge_from_other_cfg(self, other):
            for k, v in other.items():
                if isinstance(v, dict) and isinstance(self.get(k), CfgNode):
                    self[k].merge_from_other_cfg(v)
                else:
                    self[k] = CfgNode(v) if isinstance(v, dict) and not isinstance(v, CfgNode) else copy.deepcopy(v)

        @staticmethod
        def load_yaml_with_base(filename: str) -> dict:
            import yaml

            class _Loader(yaml.SafeLoader):
                pass

            def _apply_eval(loader, node):
                # the reference's YAMLs use `!!python/object/apply:eval ["[int(x * 0.1 * 384) for x in range(5, 21)]"]`
                # (configs/cityscapes/swin/unified_encoder_cityscapes.yaml:40).  Only arithmetic list comprehensions over
                # int / float / range pass: the syntax tree is checked node by node BEFORE evaluation (empty builtins alone
                # are no sandbox: attribute walks from a literal reach arbitrary code), anything else is refused.
                args = loader.construct_sequence(node)
                tree = ast.parse(args[0], mode="eval")
                allowed = (ast.Expression, ast.ListComp, ast.comprehension, ast.BinOp, ast.UnaryOp, ast.Add, ast.Sub, ast.Mult,
                           ast.Div, ast.FloorDiv, ast.Mod, ast.Pow, ast.USub, ast.UAdd, ast.Constant, ast.Name, ast.Load, ast.Store,
                           ast.Call, ast.List, ast.Tuple)
                for n in ast.walk(tree):
                    ok = isinstance(n, allowed)
                    if isinstance(n, ast.Call):
                        ok = isinstance(n.func, ast.Name) and n.func.id in ("int", "float", "range") and not n.keywords
                    if isinstance(n, ast.Constant):
                        ok = isinstance(n.value, (int, float))
                    if not ok:
                        raise ValueError(f"config expression {args[0]!r}: {type(n).__name__} is not allowed")
                return eval(compile(tree, "<cfg>", "eval"), {"__builtins__": {}, "int": int, "range": range, "float": float})

            _Loader.add_constructor("tag:yaml.org,2002:python/object/apply:eval", _apply_eval)
            with open(filename) as f:
                cfg = yaml.load(f, Loader=_Loader) or {}
            base = cfg.pop("_BASE_", None)
            if base is not None:
                if not os.path.isabs(base):
                    base = os.path.join(os.path.dirname(filename), base)
                merged = CfgNode.load_yaml_with_base(base)

                def merge(a, b):
                    for k, v in b.items():
                        if isinstance(v, dict) and isinstance(a.get(k), dict):
                            merge(a[k], v)
                        else:
                            a[k] = v
                merge(merged, cfg)
                return merged
            return cfg

        def merge_from_file(self, filename: str, allow_unsafe: bool = True):
            self.merge_from_other_cfg(CfgNode.load_yaml_with_base(filename))

        def merge_from_list(self, opts: List[Any]):
            assert len(opts) % 2 == 0
            for key, val in zip(opts[0::2], opts[1::2]):
                node = self
                parts = key.split(".")
                for p in parts[:-1]:
                    node = node[p]
                if isinstance(val, str):
                    try:
                        val = ast.literal_eval(val)
                    except (ValueError, SyntaxError):
                        pass
                node[parts[-1]] = val

    def get_cfg() -> "CfgNode":
        """Defaults for the keys of detectron2.config.defaults that the hot path or its YAMLs touch."""
        C = CfgNode
        cfg = C()
        cfg.VERSION = 2
        cfg.MODEL = C({
            "DEVICE": "cuda", "META_ARCHITECTURE": "GeneralizedRCNN", "WEIGHTS": "", "MASK_ON": False,
            "PIXEL_MEAN": [103.530, 116.280, 123.675], "PIXEL_STD": [1.0, 1.0, 1.0],
            "BACKBONE": {"NAME": "build_resnet_backbone", "FREEZE_AT": 2},
            "RESNETS": {"DEPTH": 50, "OUT_FEATURES": ["res4"], "NORM": "FrozenBN", "STEM_OUT_CHANNELS": 64,
                        "RES2_OUT_CHANNELS": 256, "STRIDE_IN_1X1": True, "RES5_MULTI_GRID": [1, 2, 4],
                        "STEM_TYPE": "basic", "NUM_GROUPS": 1, "WIDTH_PER_GROUP": 64, "RES5_DILATION": 1},
            "SEM_SEG_HEAD": {"NAME": "SemSegFPNHead", "IN_FEATURES": ["p2", "p3", "p4", "p5"], "IGNORE_VALUE": 255,
                             "NUM_CLASSES": 54, "CONVS_DIM": 128, "COMMON_STRIDE": 4, "NORM": "GN", "LOSS_WEIGHT": 1.0},
        })
        cfg.INPUT = C({"MIN_SIZE_TRAIN": (800,), "MAX_SIZE_TRAIN": 1333, "MIN_SIZE_TEST": 800, "MAX_SIZE_TEST": 1333,
                       "FORMAT": "BGR", "MASK_FORMAT": "polygon", "RANDOM_FLIP": "horizontal",
                       "CROP": {"ENABLED": False, "TYPE": "relative_range", "SIZE": [0.9, 0.9]}})
        cfg.DATASETS = C({"TRAIN": (), "TEST": ()})
        cfg.DATALOADER = C({"NUM_WORKERS": 4, "FILTER_EMPTY_ANNOTATIONS": True, "SAMPLER_TRAIN": "TrainingSampler",
                            "ASPECT_RATIO_GROUPING": True})
        cfg.SOLVER = C({"IMS_PER_BATCH": 16, "BASE_LR": 0.001, "MAX_ITER": 40000, "WEIGHT_DECAY": 0.0001,
                        "WARMUP_FACTOR": 0.001, "WARMUP_ITERS": 1000, "LR_SCHEDULER_NAME": "WarmupMultiStepLR",
                        "CLIP_GRADIENTS": {"ENABLED": False, "CLIP_TYPE": "value", "CLIP_VALUE": 1.0, "NORM_TYPE": 2.0},
                        "AMP": {"ENABLED": False}, "CHECKPOINT_PERIOD": 5000})
        cfg.TEST = C({"EVAL_PERIOD": 0, "DETECTIONS_PER_IMAGE": 100,
                      "AUG": {"ENABLED": False, "MIN_SIZES": (400, 500, 600), "MAX_SIZE": 4000, "FLIP": True}})
        cfg.OUTPUT_DIR = "./output"
        cfg.SEED = -1
        return cfg

    class ShapeSpec:
        def __init__(self, channels=None, height=None, width=None, stride=None):
            self.channels, self.height, self.width, self.stride = channels, height, width, stride

        def __repr__(self):
            return f"ShapeSpec(channels={self.channels}, height={self.height}, width={self.width}, stride={self.stride})"

    def configurable(init_func=None, *, from_config=None):
        """`@configurable` on __init__: `Cls(cfg, *a)` is routed through `Cls.from_config(cfg, *a)`."""
        assert init_func is not None and inspect.isfunction(init_func) and init_func.__name__ == "__init__"

        @functools.wraps(init_func)
        def wrapped(self, *args, **kwargs):
            from_cfg = type(self).from_config
            if (len(args) and isinstance(args[0], CfgNode)) or isinstance(kwargs.get("cfg"), CfgNode):
                explicit = from_cfg(*args, **kwargs)
                init_func(self, **explicit)
            else:
                init_func(self, *args, **kwargs)
        return wrapped

    class Backbone(nn.Module):
        @property
        def size_divisibility(self) -> int:
            return 0

        def output_shape(self):
            return {name: ShapeSpec(channels=self._out_feature_channels[name], stride=self._out_feature_strides[name])
                    for name in self._out_features}

    class Boxes:
        """detectron2.structures.Boxes, the slice instance_inference uses: an (N, 4) tensor holder."""

        def __init__(self, tensor: torch.Tensor):
            self.tensor = tensor

        def __len__(self):
            return self.tensor.shape[0]

    class Instances:
        """detectron2.structures.Instances, the slice instance_inference uses: per-image fields set as attributes."""

        def __init__(self, image_size: Tuple[int, int], **kwargs):
            object.__setattr__(self, "_image_size", tuple(image_size))
            object.__setattr__(self, "_fields", {})
            for k, v in kwargs.items():
                setattr(self, k, v)

        @property
        def image_size(self):
            return self._image_size

        def __setattr__(self, name, val):
            self._fields[name] = val

        def __getattr__(self, name):
            if name == "_fields" or name not in self._fields:
                raise AttributeError(f"Cannot find field '{name}' in the given Instances!")
            return self._fields[name]

        def has(self, name):
            return name in self._fields

        def get_fields(self):
            return self._fields

        def __len__(self):
            for v in self._fields.values():
                return len(v)
            raise NotImplementedError("Empty Instances does not support __len__!")

    class ImageList:
        def __init__(self, tensor: torch.Tensor, image_sizes: List[Tuple[int, int]]):
            self.tensor, self.image_sizes = tensor, image_sizes

        def __len__(self):
            return len(self.image_sizes)

        @staticmethod
        def from_tensors(tensors: List[torch.Tensor], size_divisibility: int = 0, pad_value: float = 0.0) -> "ImageList":
            sizes = [(t.shape[-2], t.shape[-1]) for t in tensors]
            H, W = max(s[0] for s in sizes), max(s[1] for s in sizes)
            if size_divisibility > 1:
                d = size_divisibility
                H, W = (H + d - 1) // d * d, (W + d - 1) // d * d
            out = tensors[0].new_full((len(tensors),) + tuple(tensors[0].shape[:-2]) + (H, W), pad_value)
            for i, t in enumerate(tensors):
                out[i, ..., : t.shape[-2], : t.shape[-1]].copy_(t)
            return ImageList(out.contiguous(), sizes)

    class Conv2d(nn.Conv2d):
        def __init__(self, *args, **kwargs):
            norm = kwargs.pop("norm", None)
            activation = kwargs.pop("activation", None)
            super().__init__(*args, **kwargs)
            self.norm, self.activation = norm, activation

        def forward(self, x):
            x = F.conv2d(x, self.weight, self.bias, self.stride, self.padding, self.dilation, self.groups)
            if self.norm is not None:
                x = self.norm(x)
            if self.activation is not None:
                x = self.activation(x)
            return x

    def get_norm(norm, out_channels):
        if norm is None or (isinstance(norm, str) and len(norm) == 0):
            return None
        if isinstance(norm, str):
            if norm == "GN":
                return nn.GroupNorm(32, out_channels)
            if norm == "LN":     # detectron2's "LN" is a per-pixel channel LayerNorm, NOT GroupNorm(1, C); no shipped config uses it
                raise NotImplementedError("norm 'LN' (detectron2 channel LayerNorm) is outside the hot path (SURVEY.md §8)")
            raise NotImplementedError(f"norm {norm!r} is outside the hot path (SURVEY.md §8)")
        return norm(out_channels)

    BACKBONE_REGISTRY = Registry("BACKBONE")
    SEM_SEG_HEADS_REGISTRY = Registry("SEM_SEG_HEADS")
    META_ARCH_REGISTRY = Registry("META_ARCH")

    def build_backbone(cfg, input_shape=None):
        if input_shape is None:
            input_shape = ShapeSpec(channels=len(cfg.MODEL.PIXEL_MEAN))
        return BACKBONE_REGISTRY.get(cfg.MODEL.BACKBONE.NAME)(cfg, input_shape)

    def build_sem_seg_head(cfg, input_shape):
        return SEM_SEG_HEADS_REGISTRY.get(cfg.MODEL.SEM_SEG_HEAD.NAME)(cfg, input_shape)

    def build_model(cfg):
        model = META_ARCH_REGISTRY.get(cfg.MODEL.META_ARCHITECTURE)(cfg)
        model.to(torch.device(cfg.MODEL.DEVICE))
        return model
