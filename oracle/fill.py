"""Deterministic name-hashed parameter filler.

TEST INFRASTRUCTURE (oracle side).  No checkpoint ships with the reference
(`configs/cityscapes/Base-Cityscapes-UnifiedSegmentation.yaml:5` points at the
authors' disk), so parity is pinned on synthetic weights: the reference modules
(imported in the build container only), the CPU restatement in
`oracle/torch_ref.py` and the HIP product all receive the *same* tensors,
generated from nothing but the parameter's state-dict name and shape.

The distributions are chosen so every code path carries signal (non-trivial
LayerNorm affine, non-zero biases, peaky-enough softmax, sampling offsets of a
few pixels) while activations stay O(1) through 24+ residual blocks.
"""
import hashlib

import numpy as np
import torch


def _rng(name: str) -> np.random.Generator:
    seed = int.from_bytes(hashlib.sha256(name.encode()).digest()[:8], "little")
    return np.random.Generator(np.random.PCG64(seed))


def tensor_for(name: str, shape) -> torch.Tensor:
    """fp32 tensor for state-dict entry `name` of shape `shape`."""
    shape = tuple(int(s) for s in shape)
    rng = _rng(name)
    leaf = name.split(".")[-1]
    n = rng.standard_normal(shape, dtype=np.float64)
    if len(shape) >= 2:
        fan_in = int(np.prod(shape[1:]))
        std = fan_in ** -0.5
        if leaf == "relative_position_bias_table" or leaf == "rpb":
            std = 0.5
        elif "embed" in name and leaf == "weight" and len(shape) == 2 and "mask_embed" not in name \
                and "class_embed" not in name and "patch_embed" not in name:
            std = 0.5  # nn.Embedding tables (query_embed, level_embed)
        elif leaf == "level_embed":
            std = 0.5
        elif "qkv" in name or "in_proj_weight" in name:
            std = 1.5 * fan_in ** -0.5  # sharper attention logits
        out = n * std
    else:
        if leaf == "running_var":  # BatchNorm running variance: positive, O(1)
            out = 0.6 + 0.4 * np.abs(n)
        elif leaf == "num_batches_tracked":
            return torch.zeros(shape, dtype=torch.int64)
        elif leaf == "weight":  # LayerNorm / GroupNorm / BatchNorm scale
            out = 1.0 + 0.1 * n
        elif "sampling_offsets" in name:
            out = 2.0 * n  # a few pixels of spread, like the reference's grid init
        else:
            out = 0.1 * n
    return torch.from_numpy(out.astype(np.float32))


@torch.no_grad()
def fill_module(module: torch.nn.Module, prefix: str = "") -> None:
    """Overwrite every floating-point parameter of `module` in place, and the running statistics of its BatchNorm layers (the
    sequence-branch decoders run them in eval mode).  Other buffers (e.g. `relative_position_index`) keep their constructed values.
    """
    for name, p in module.named_parameters():
        p.copy_(tensor_for(prefix + name, p.shape))
    for name, b in module.named_buffers():
        if name.endswith(("running_mean", "running_var")):
            b.copy_(tensor_for(prefix + name, b.shape))


def state_dict_for(shapes: dict, prefix: str = "") -> dict:
    """{name: shape} -> {name: tensor}."""
    return {k: tensor_for(prefix + k, s) for k, s in shapes.items()}
