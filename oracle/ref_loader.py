"""Import the reference's own PyTorch modules for the hot path, on CPU.

TEST INFRASTRUCTURE — build container only.  `/root/reference` is never copied
and does not exist on the GPU box; everything here skips cleanly when it is
absent.  The reference needs third-party packages that are not installed
(detectron2, timm, fvcore, natten); the handful of symbols the hot-path files
touch are provided as in-process stand-ins with the documented semantics
(SURVEY.md §8c).  Nothing from the reference's files is restated here.

Loaded files (all under /root/reference/model/modeling):
  backbone/swin.py
  pixel_decoder/msdeformattn.py  (+ ops/modules, ops/functions: CPU grid_sample branch)
  transformer_decoder/{oneformer_transformer_decoder,transformer,position_encoding}.py
"""
import importlib.util
import os
import sys
import types

import torch
import torch.nn as nn
import torch.nn.functional as F

REF_ROOT = os.environ.get("UENC_REFERENCE_ROOT", "/root/reference")


def available() -> bool:
    return os.path.isdir(os.path.join(REF_ROOT, "model", "modeling"))


class _ShapeSpec:
    def __init__(self, channels=None, height=None, width=None, stride=None):
        self.channels, self.height, self.width, self.stride = channels, height, width, stride


class _Registry:
    def __init__(self, name):
        self._name, self._map = name, {}

    def register(self, obj=None):
        if obj is None:
            def deco(o):
                self._map[o.__name__] = o
                return o
            return deco
        self._map[obj.__name__] = obj
        return obj

    def get(self, name):
        return self._map[name]


class _DropPath(nn.Module):
    """timm DropPath: per-sample Bernoulli keep, scaled by 1/keep (train only)."""

    def __init__(self, drop_prob=0.0):
        super().__init__()
        self.drop_prob = drop_prob

    def forward(self, x):
        if self.drop_prob == 0.0 or not self.training:
            return x
        keep = 1 - self.drop_prob
        shape = (x.shape[0],) + (1,) * (x.ndim - 1)
        mask = x.new_empty(shape).bernoulli_(keep)
        return x * mask / keep


class _Conv2d(nn.Conv2d):
    """detectron2.layers.Conv2d: conv -> optional norm -> optional activation."""

    def __init__(self, *args, **kwargs):
        norm = kwargs.pop("norm", None)
        activation = kwargs.pop("activation", None)
        super().__init__(*args, **kwargs)
        self.norm = norm
        self.activation = activation

    def forward(self, x):
        x = F.conv2d(x, self.weight, self.bias, self.stride, self.padding, self.dilation, self.groups)
        if self.norm is not None:
            x = self.norm(x)
        if self.activation is not None:
            x = self.activation(x)
        return x


def _get_norm(norm, out_channels):
    if norm is None or norm == "":
        return None
    if norm == "GN":
        return nn.GroupNorm(32, out_channels)
    raise NotImplementedError(norm)


def _configurable(init_func=None, *, from_config=None):
    # explicit-kwargs construction only: the decorator is the identity
    if init_func is not None:
        return init_func
    return lambda f: f


def _mod(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


_installed = False
_saved = {}


def _install_stubs():
    global _installed
    if _installed:
        return
    to_2tuple = lambda x: tuple(x) if isinstance(x, (tuple, list)) else (x, x)
    trunc_normal_ = lambda t, std=1.0, **kw: nn.init.trunc_normal_(t, std=std)
    _mod("timm")
    _mod("timm.models")
    _mod("timm.models.layers", DropPath=_DropPath, to_2tuple=to_2tuple, trunc_normal_=trunc_normal_)
    backbone_reg, seg_reg = _Registry("BACKBONE"), _Registry("SEM_SEG_HEADS")
    _mod("detectron2")
    _mod("detectron2.modeling", BACKBONE_REGISTRY=backbone_reg, Backbone=nn.Module,
         ShapeSpec=_ShapeSpec, SEM_SEG_HEADS_REGISTRY=seg_reg)
    _mod("detectron2.config", configurable=_configurable)
    _mod("detectron2.layers", Conv2d=_Conv2d, ShapeSpec=_ShapeSpec, get_norm=_get_norm, DeformConv=None)
    _mod("detectron2.utils")
    _mod("detectron2.utils.registry", Registry=_Registry)
    _mod("fvcore")
    _mod("fvcore.nn")
    _mod("fvcore.nn.weight_init", c2_xavier_fill=lambda m: None, c2_msra_fill=lambda m: None)
    sys.modules["fvcore.nn"].weight_init = sys.modules["fvcore.nn.weight_init"]
    # package shells whose __path__ points into the reference, bypassing the
    # __init__.py files that drag in natten / datasets / evaluation.
    base = os.path.join(REF_ROOT, "model")
    for name, sub in [
        ("model", ""),
        ("model.modeling", "modeling"),
        ("model.modeling.backbone", "modeling/backbone"),
        ("model.modeling.transformer_decoder", "modeling/transformer_decoder"),
        ("model.modeling.pixel_decoder", "modeling/pixel_decoder"),
        ("model.modeling.pixel_decoder.ops", "modeling/pixel_decoder/ops"),
        ("model.modeling.pixel_decoder.ops.functions", "modeling/pixel_decoder/ops/functions"),
        ("model.modeling.pixel_decoder.ops.modules", "modeling/pixel_decoder/ops/modules"),
    ]:
        _saved[name] = sys.modules.get(name)
        m = types.ModuleType(name)
        m.__path__ = [os.path.join(base, sub)]
        sys.modules[name] = m
    _installed = True


def _load(modname, relpath):
    if modname in sys.modules and getattr(sys.modules[modname], "__file__", None):
        return sys.modules[modname]
    spec = importlib.util.spec_from_file_location(modname, os.path.join(REF_ROOT, "model", relpath))
    m = importlib.util.module_from_spec(spec)
    sys.modules[modname] = m
    spec.loader.exec_module(m)
    return m


def load():
    """Returns a namespace with the reference classes used by the goldens."""
    assert available(), "reference tree not present"
    _install_stubs()
    ns = types.SimpleNamespace()
    swin = _load("model.modeling.backbone.swin", "modeling/backbone/swin.py")
    func = _load("model.modeling.pixel_decoder.ops.functions.ms_deform_attn_func",
                 "modeling/pixel_decoder/ops/functions/ms_deform_attn_func.py")
    sys.modules["model.modeling.pixel_decoder.ops.functions"].ms_deform_attn_func = func
    mods = _load("model.modeling.pixel_decoder.ops.modules.ms_deform_attn",
                 "modeling/pixel_decoder/ops/modules/ms_deform_attn.py")
    sys.modules["model.modeling.pixel_decoder.ops.modules"].MSDeformAttn = mods.MSDeformAttn
    pe = _load("model.modeling.transformer_decoder.position_encoding",
               "modeling/transformer_decoder/position_encoding.py")
    tr = _load("model.modeling.transformer_decoder.transformer",
               "modeling/transformer_decoder/transformer.py")
    pix = _load("model.modeling.pixel_decoder.msdeformattn", "modeling/pixel_decoder/msdeformattn.py")
    dec = _load("model.modeling.transformer_decoder.oneformer_transformer_decoder",
                "modeling/transformer_decoder/oneformer_transformer_decoder.py")
    ns.swin, ns.msda_func, ns.msda_mod, ns.pe, ns.transformer, ns.pixdec, ns.dec = swin, func, mods, pe, tr, pix, dec
    ns.ShapeSpec = _ShapeSpec
    return ns


def load_meta_arch():
    """The reference's `OneFormer` class object (model/oneformer_model.py), for its pure post-processing methods
    (`semantic_inference`, `panoptic_inference`, `instance_inference`: oneformer_model.py:367-489), called unbound on a
    SimpleNamespace `self`.  The file's module-level imports need more of detectron2 (stand-ins below, documented semantics)
    and three of the reference's own modules that only the depth / pose / motion branch uses; those are satisfied by empty
    name holders so that their files (and their further dependencies) are never executed."""
    assert available(), "reference tree not present"
    ns = load()

    def sem_seg_postprocess(result, img_size, output_height, output_width):
        # detectron2.modeling.postprocessing.sem_seg_postprocess: crop the padding away, resize to the requested resolution
        result = result[:, : img_size[0], : img_size[1]].expand(1, -1, -1, -1)
        return F.interpolate(result, size=(output_height, output_width), mode="bilinear", align_corners=False)[0]

    class _Instances:
        def __init__(self, image_size):
            self.image_size = image_size

    class _Boxes:
        def __init__(self, tensor):
            self.tensor = tensor

    d2m = sys.modules["detectron2.modeling"]
    d2m.META_ARCH_REGISTRY = _Registry("META_ARCH")
    d2m.build_backbone = d2m.build_sem_seg_head = None
    _mod("detectron2.data", MetadataCatalog=None)
    _mod("detectron2.modeling.backbone", Backbone=nn.Module)
    _mod("detectron2.modeling.postprocessing", sem_seg_postprocess=sem_seg_postprocess)
    _mod("detectron2.structures", Boxes=_Boxes, ImageList=None, Instances=_Instances, BitMasks=None)
    _mod("detectron2.utils.memory", retry_if_cuda_oom=lambda f: f)
    base = os.path.join(REF_ROOT, "model")
    for name, sub in [("model.modeling.motion_decoder", "modeling/motion_decoder"), ("model.modeling.pose_decoder", "modeling/pose_decoder"),
                      ("model.data", "data")]:
        m = types.ModuleType(name)
        m.__path__ = [os.path.join(base, sub)]
        sys.modules[name] = m
    _mod("model.modeling.motion_decoder.dynamo_motion_decoder_mod", MotionDecoderV2=None)
    _mod("model.modeling.pose_decoder.resnet_like_pose_decoder", ResNetLike=None)
    _mod("model.modeling.monodepth_loss", transformation_from_parameters=None)
    _mod("model.data.tokenizer", SimpleTokenizer=None, Tokenize=None)
    m = _load("model.oneformer_model", "oneformer_model.py")
    ns.OneFormer = m.OneFormer
    return ns


def load_sequence():
    """The reference's "sequence"-branch modules (pose_decoder/resnet_like_pose_decoder.py, motion_decoder/dynamo_motion_decoder_mod.py,
    pixel_decoder/transdssl.py) and `transformation_from_parameters` of monodepth_loss.py.  The first two import nothing but torch;
    transdssl.py needs the registry / ShapeSpec stand-ins already installed; monodepth_loss.py imports cv2 (absent) and a
    distributed helper at module level that the pose functions never touch: both are satisfied by empty name holders."""
    assert available(), "reference tree not present"
    _install_stubs()
    base = os.path.join(REF_ROOT, "model")
    for name, sub in [("model.modeling.motion_decoder", "modeling/motion_decoder"), ("model.modeling.pose_decoder", "modeling/pose_decoder"),
                      ("model.utils", "utils")]:
        if name not in sys.modules or not getattr(sys.modules[name], "__path__", None):
            m = types.ModuleType(name)
            m.__path__ = [os.path.join(base, sub)]
            sys.modules[name] = m
    for name in ("model.modeling.motion_decoder.dynamo_motion_decoder_mod", "model.modeling.pose_decoder.resnet_like_pose_decoder",
                 "model.modeling.monodepth_loss"):
        m = sys.modules.get(name)
        if m is not None and getattr(m, "__file__", None) is None:      # name holders left by load_meta_arch()
            del sys.modules[name]
    if "cv2" not in sys.modules:
        _mod("cv2")
    _mod("model.utils.misc", is_dist_avail_and_initialized=lambda: False)
    ns = types.SimpleNamespace()
    ns.pose = _load("model.modeling.pose_decoder.resnet_like_pose_decoder", "modeling/pose_decoder/resnet_like_pose_decoder.py")
    ns.motion = _load("model.modeling.motion_decoder.dynamo_motion_decoder_mod", "modeling/motion_decoder/dynamo_motion_decoder_mod.py")
    ns.transdssl = _load("model.modeling.pixel_decoder.transdssl", "modeling/pixel_decoder/transdssl.py")
    ns.geometry = _load("model.modeling.monodepth_loss", "modeling/monodepth_loss.py")
    ns.ShapeSpec = _ShapeSpec
    return ns


def load_data_eval():
    """The reference's evaluation loop, depth metrics, KITTI ground-truth projection and dataset registration, for the fixtures of
    SURVEY.md §8f rank 4 (oracle/make_data_eval_golden.py).  Files executed: model/evaluation/evaluator.py,
    model/evaluation/kitti_evaluation.py, model/data/datasets/register_cityscapes_panoptic.py, model/data/datasets/register_kitti.py,
    model/modeling/monodepth_loss.py (for disp_to_depth).  Stand-ins for what they import and the image lacks -- documented
    semantics only: detectron2.utils.comm (single process), detectron2.utils.logger.log_every_n_seconds (a plain log call),
    detectron2.data.{DatasetCatalog, MetadataCatalog} (name -> function / attribute bag), detectron2.utils.file_io.PathManager
    (os / open), detectron2.data.datasets.builtin_meta.CITYSCAPES_CATEGORIES (the public 19-class label table [not in reference]),
    detectron2.utils.events.get_event_storage (raises, as outside a trainer); cv2 / matplotlib.cm / skimage are imported by
    kitti_evaluation.py at module level but not touched by the functions called here: empty name holders.  numpy 2 removed the
    `np.int` alias the reference's generate_depth_map uses (kitti_evaluation.py:152): restored as `int` while the fixture is made."""
    assert available(), "reference tree not present"
    _install_stubs()
    import logging
    import numpy as np

    class _Comm:
        get_world_size = staticmethod(lambda: 1)
        get_local_size = staticmethod(lambda: 1)
        get_rank = staticmethod(lambda: 0)
        is_main_process = staticmethod(lambda: True)
        synchronize = staticmethod(lambda: None)
        all_gather = staticmethod(lambda x: [x])

    comm = _mod("detectron2.utils.comm", **{k: getattr(_Comm, k) for k in ("get_world_size", "get_local_size", "get_rank", "is_main_process", "synchronize", "all_gather")})
    sys.modules["detectron2.utils"].comm = comm

    def log_every_n_seconds(lvl, msg, n=1, *, name=None):
        logging.getLogger(name or "model.evaluation.evaluator").log(lvl, msg)
    _mod("detectron2.utils.logger", log_every_n_seconds=log_every_n_seconds)

    class _Meta:
        def __init__(self, name):
            self.name = name

        def set(self, **kw):
            self.__dict__.update(kw)
            return self

    class _MetaCat(dict):
        def get(self, name):
            return self.setdefault(name, _Meta(name))

    class _DataCat(dict):
        def list(self):
            return list(self.keys())

        def remove(self, name):
            self.pop(name)

        def register(self, name, func):
            self[name] = func

        def get(self, name):
            return self[name]()

    class _PathManager:
        ls = staticmethod(lambda d: sorted(os.listdir(d)))
        isfile = staticmethod(os.path.isfile)
        open = staticmethod(open)

    from uenc.datasets import CITYSCAPES_CATEGORIES     # the public label table (detectron2 builtin_meta) -- not reference content
    ns = types.SimpleNamespace(DatasetCatalog=_DataCat(), MetadataCatalog=_MetaCat())
    _mod("detectron2.data", DatasetCatalog=ns.DatasetCatalog, MetadataCatalog=ns.MetadataCatalog)
    _mod("detectron2.data.datasets")
    _mod("detectron2.data.datasets.builtin_meta", CITYSCAPES_CATEGORIES=CITYSCAPES_CATEGORIES)
    _mod("detectron2.utils.file_io", PathManager=_PathManager)

    def get_event_storage():
        raise AssertionError("get_event_storage() has to be called inside a 'with EventStorage(...)' context!")
    _mod("detectron2.utils.events", get_event_storage=get_event_storage)
    for name in ("cv2", "skimage", "matplotlib"):
        if name not in sys.modules:
            _mod(name)
    if "matplotlib.cm" not in sys.modules:
        sys.modules["matplotlib"].cm = _mod("matplotlib.cm")
    if not hasattr(np, "int"):
        np.int = int
    base = os.path.join(REF_ROOT, "model")
    for name, sub in [("model.evaluation", "evaluation"), ("model.data", "data"), ("model.data.datasets", "data/datasets"), ("model.utils", "utils")]:
        m = sys.modules.get(name)
        if m is None or not getattr(m, "__path__", None) or getattr(m, "__file__", None):
            m = types.ModuleType(name)
            m.__path__ = [os.path.join(base, sub)]
            sys.modules[name] = m
    if "model.utils.misc" not in sys.modules:
        _mod("model.utils.misc", is_dist_avail_and_initialized=lambda: False)
    m = sys.modules.get("model.modeling.monodepth_loss")
    if m is not None and getattr(m, "__file__", None) is None:
        del sys.modules["model.modeling.monodepth_loss"]
    ns.geometry = _load("model.modeling.monodepth_loss", "modeling/monodepth_loss.py")
    ns.evaluator = _load("model.evaluation.evaluator", "evaluation/evaluator.py")
    ns.kitti_eval = _load("model.evaluation.kitti_evaluation", "evaluation/kitti_evaluation.py")
    ns.reg_cityscapes = _load("model.data.datasets.register_cityscapes_panoptic", "data/datasets/register_cityscapes_panoptic.py")
    ns.reg_kitti = _load("model.data.datasets.register_kitti", "data/datasets/register_kitti.py")
    return ns
