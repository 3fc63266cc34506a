"""Checkpoint ingest for the hot path: reference checkpoints load into this model unchanged.

What the reference uses (all through Detectron2, none of it in /root/reference as code):
  * `DetectionCheckpointer(model).resume_or_load(cfg.MODEL.WEIGHTS)` (train_net.py:287) / `.load(cfg.MODEL.WEIGHTS)`
    (demo/defaults.py:56-57) on `.pth` files (torch.save) and on `.pkl` wrapper files
    `{"model": state_dict, "__author__": str, "matching_heuristics": bool}` written by
    tools/convert-pretrained-model-to-d2.py:22-30, tools/convert-pretrained-nat-model-to-d2.py:22-30 and
    tools/merge_two_pretrained_models.py:19-33;
  * with `matching_heuristics` the checkpoint's keys are matched to the model's by longest common suffix (a raw Swin
    checkpoint's `patch_embed.proj.weight` lands on `backbone.patch_embed.proj.weight`), otherwise by exact name;
  * the two `_load_from_state_dict` key-renaming shims (meta_arch/oneformer_head.py:26-48,
    transformer_decoder/oneformer_transformer_decoder.py:231-252) -- they live on the modules (`uenc/modeling/...`).

Files are read WITHOUT executing anything they contain: `.pth` through `torch.load(weights_only=True)`, wrapper pickles
through an unpickler that resolves only the handful of constructors a tensor / ndarray state dict needs (anything else in
the stream raises).  The reference's tools unpickle with plain `pickle.load`; a checkpoint is data and is treated as such here.
Written wrappers hold numpy arrays (what Detectron2's own model-zoo `.pkl` files hold; it converts them on load).
"""
import collections
import io
import logging
import os
import pickle
from typing import Dict, List, Optional, Tuple

import numpy as np
import torch
import torch.nn as nn

logger = logging.getLogger(__name__)


# ---------------------------------------------------------------------------------------------------------------------
# reading without executing
# ---------------------------------------------------------------------------------------------------------------------
def _storage_from_bytes(b: bytes):
    """Replacement for torch.storage._load_from_bytes (what a tensor inside a PLAIN pickle reduces to): the nested
    torch-serialised storage is read with the weights-only loader instead of an unrestricted one."""
    return torch.load(io.BytesIO(b), map_location="cpu", weights_only=True)


def _latin1_encode(s, encoding="latin1"):
    """What numpy array bytes pickle to under protocol <= 2: `_codecs.encode(<str>, "latin1")`.  Only that exact use is honoured."""
    if not isinstance(s, str) or encoding.lower().replace("-", "") != "latin1":
        raise pickle.UnpicklingError("checkpoint uses _codecs.encode for something other than latin1 array bytes, refused")
    return s.encode("latin1")


class _WeightsOnlyUnpickler(pickle.Unpickler):
    """Resolves exactly what a state dict of tensors / ndarrays pickles to; every other global is refused."""

    def find_class(self, module, name):
        if (module, name) == ("torch.storage", "_load_from_bytes"):
            return _storage_from_bytes
        if (module, name) == ("collections", "OrderedDict"):
            return collections.OrderedDict
        if (module, name) == ("_codecs", "encode"):
            return _latin1_encode
        if (module, name) in (("torch._utils", "_rebuild_tensor_v2"), ("torch._utils", "_rebuild_parameter")):
            import torch._utils as tu
            return getattr(tu, name)
        if module == "torch" and name in ("Size", "float32", "float16", "bfloat16", "float64", "int64", "int32", "uint8", "bool",
                                           "FloatStorage", "HalfStorage", "BFloat16Storage", "DoubleStorage", "LongStorage",
                                           "IntStorage", "ByteStorage", "BoolStorage"):
            return getattr(torch, name)
        if module in ("numpy.core.multiarray", "numpy._core.multiarray") and name in ("_reconstruct", "scalar"):
            try:
                import numpy._core.multiarray as ma
            except ImportError:                       # numpy < 2
                import numpy.core.multiarray as ma
            return getattr(ma, name)
        if (module, name) in (("numpy", "ndarray"), ("numpy", "dtype")):
            return getattr(np, name)
        raise pickle.UnpicklingError(f"checkpoint refers to {module}.{name}: not a weights-only pickle, refused")


def _to_tensor_dict(sd) -> "collections.OrderedDict[str, torch.Tensor]":
    out = collections.OrderedDict()
    for k, v in sd.items():
        if isinstance(v, np.ndarray):
            v = torch.from_numpy(v if v.flags.c_contiguous else np.ascontiguousarray(v))   # (ascontiguousarray would turn a 0-d array into (1,))
        elif isinstance(v, torch.nn.Parameter):
            v = v.data
        if not torch.is_tensor(v):
            raise ValueError(f"checkpoint entry {k!r} is a {type(v).__name__}, not an array")
        out[str(k)] = v
    return out


def read_checkpoint(path: str) -> dict:
    """-> {"model": OrderedDict name -> CPU tensor, "__author__": str | None, "matching_heuristics": bool}.

    `.pkl`: the Detectron2 wrapper (or a bare state dict, as convert-pretrained-nat-model-to-d2.py's INPUT may be);
    anything else: a torch.save file holding a state dict or `{"model": state_dict, ...}`."""
    if path.endswith(".pkl"):
        with open(path, "rb") as f:
            data = _WeightsOnlyUnpickler(f, encoding="latin1").load()
    else:
        data = torch.load(path, map_location="cpu", weights_only=True)
    if not isinstance(data, dict):
        raise ValueError(f"{path}: expected a dict, got {type(data).__name__}")
    if "model" in data and isinstance(data["model"], dict):
        meta = {"__author__": data.get("__author__"), "matching_heuristics": bool(data.get("matching_heuristics", False))}
        sd = data["model"]
    else:
        meta = {"__author__": None, "matching_heuristics": False}
        sd = data
    return {"model": _to_tensor_dict(sd), **meta}


def write_wrapper(path: str, state_dict: Dict[str, torch.Tensor], author: str = "third_party", matching_heuristics: bool = False):
    """The `.pkl` wrapper of tools/convert-pretrained-model-to-d2.py:26-30, arrays stored as numpy (bf16 as float32)."""
    model = collections.OrderedDict()
    for k, v in state_dict.items():
        t = v.detach().cpu() if torch.is_tensor(v) else torch.as_tensor(v)
        model[k] = (t.float() if t.dtype == torch.bfloat16 else t).numpy()
    with open(path, "wb") as f:
        pickle.dump({"model": model, "__author__": author, "matching_heuristics": bool(matching_heuristics)}, f, protocol=2)


# ---------------------------------------------------------------------------------------------------------------------
# the three tools (same file-level behaviour as the reference's scripts; CLIs in uni-encoder-code_amd/tools/)
# ---------------------------------------------------------------------------------------------------------------------
def convert_pretrained_model_to_d2(src: str, dst: str):
    """tools/convert-pretrained-model-to-d2.py: `torch.load(src)["model"]` -> wrapper, exact-name matching."""
    ck = read_checkpoint(src)
    write_wrapper(dst, ck["model"], "third_party", False)


def convert_pretrained_nat_model_to_d2(src: str, dst: str):
    """tools/convert-pretrained-nat-model-to-d2.py: the WHOLE loaded object is the state dict, suffix matching on."""
    data = torch.load(src, map_location="cpu", weights_only=True)
    if not isinstance(data, dict):
        raise ValueError(f"{src}: expected a state dict")
    write_wrapper(dst, _to_tensor_dict(data), "third_party", True)


def merge_two_pretrained_models(a: str, b: str, dst: str):
    """tools/merge_two_pretrained_models.py: model = a["model"] updated with b["model"] (b wins on a name clash), suffix matching on."""
    m = read_checkpoint(a)["model"]
    m.update(read_checkpoint(b)["model"])
    write_wrapper(dst, m, "third_party", True)


# ---------------------------------------------------------------------------------------------------------------------
# loading into a model
# ---------------------------------------------------------------------------------------------------------------------
def align_by_suffix(model_keys: List[str], ckpt: Dict[str, torch.Tensor], model_shapes: Dict[str, Tuple[int, ...]]):
    """Detectron2's `matching_heuristics` [not in reference: detectron2/checkpoint/c2_model_loading.py, restated]: every model
    key takes the checkpoint key that is its longest suffix at a '.' boundary (or equal); a checkpoint key is used once; a
    shape mismatch drops the pair.  Returns (renamed state dict, unmatched checkpoint keys)."""
    ck = sorted(ckpt.keys(), key=len, reverse=True)
    used, out = set(), collections.OrderedDict()
    for mk in model_keys:
        for c in ck:
            if c in used:
                continue
            if mk == c or mk.endswith("." + c):
                if tuple(ckpt[c].shape) == tuple(model_shapes[mk]):
                    out[mk] = ckpt[c]
                    used.add(c)
                else:
                    logger.warning("shape mismatch for %s <- %s: %s vs %s", mk, c, tuple(model_shapes[mk]), tuple(ckpt[c].shape))
                break
    return out, [c for c in ckpt if c not in used]


class DetectionCheckpointer:
    """The slice of detectron2.checkpoint.DetectionCheckpointer the reference's drivers call (train_net.py:287,
    demo/defaults.py:56-57): load / resume_or_load / save on one model."""

    def __init__(self, model: nn.Module, save_dir: str = ""):
        self.model, self.save_dir = model, save_dir

    def load(self, path: str, checkpointables=None) -> dict:
        if not path:
            logger.info("No checkpoint found. Initializing model from scratch")
            return {}
        if not os.path.isfile(path):
            raise FileNotFoundError(f"Checkpoint {path} not found!")
        ck = read_checkpoint(path)
        sd = ck["model"]
        own = self.model.state_dict()
        if ck["matching_heuristics"]:
            sd, unmatched = align_by_suffix(list(own.keys()), sd, {k: tuple(v.shape) for k, v in own.items()})
        else:
            unmatched = []
        # shape-incompatible entries are dropped (and reported), as fvcore's Checkpointer does, instead of failing the whole load
        dropped = [k for k, v in sd.items() if k in own and tuple(own[k].shape) != tuple(v.shape)]
        for k in dropped:
            del sd[k]
        res = self.model.load_state_dict(sd, strict=False)          # runs the modules' legacy-key shims
        report = {"missing_keys": list(res.missing_keys), "unexpected_keys": list(res.unexpected_keys) + unmatched,
                  "incorrect_shapes": dropped, "matching_heuristics": ck["matching_heuristics"], "__author__": ck["__author__"]}
        if report["missing_keys"]:
            logger.warning("checkpoint %s: %d model keys not found (first: %s)", path, len(report["missing_keys"]), report["missing_keys"][:3])
        try:
            from . import ops
            ops.CACHE.invalidate()                                   # bf16 operand copies follow the new weights
        except Exception:
            pass
        return report

    def resume_or_load(self, path: str, *, resume: bool = True) -> dict:
        last = os.path.join(self.save_dir, "last_checkpoint") if self.save_dir else ""
        if resume and last and os.path.isfile(last):
            with open(last) as f:
                path = os.path.join(self.save_dir, f.read().strip())
        return self.load(path)

    def save(self, name: str, **extra) -> str:
        """`{save_dir}/{name}.pth` = torch.save({"model": state_dict, **extra}) + the `last_checkpoint` pointer file."""
        os.makedirs(self.save_dir or ".", exist_ok=True)
        fn = os.path.join(self.save_dir, name + ".pth")
        torch.save({"model": collections.OrderedDict((k, v.detach().cpu()) for k, v in self.model.state_dict().items()), **extra}, fn)
        with open(os.path.join(self.save_dir, "last_checkpoint"), "w") as f:
            f.write(name + ".pth")
        return fn
