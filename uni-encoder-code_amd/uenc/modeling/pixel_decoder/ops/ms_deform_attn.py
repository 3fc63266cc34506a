"""MSDeformAttn module + autograd Function on the HIP kernel.

Counterpart of reference model/modeling/pixel_decoder/ops/modules/ms_deform_attn.py:37-126 and
ops/functions/ms_deform_attn_func.py:35-52 (same parameter names: sampling_offsets,
attention_weights, value_proj, output_proj; same forward signature).  The four projections run as
bf16 MFMA GEMMs, the sampling core as `uenc_msdeform_attn_{fwd,bwd}`.  There is no CPU branch: the
reference's `ms_deform_attn_core_pytorch` debug path is restated only in the oracle.
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F

from .... import ops
from ....ops import MSDeformAttnFunction


def ms_deform_attn_core_pytorch(*args, **kwargs):
    raise RuntimeError("the grid_sample debug path is not part of the product; see oracle/torch_ref.py:ms_deform_attn_core")


class MSDeformAttn(nn.Module):
    def __init__(self, d_model=256, n_levels=4, n_heads=8, n_points=4):
        super().__init__()
        if d_model % n_heads != 0:
            raise ValueError("d_model must be divisible by n_heads, but got {} and {}".format(d_model, n_heads))
        self.im2col_step = 128
        self.d_model, self.n_levels, self.n_heads, self.n_points = d_model, n_levels, n_heads, n_points
        self.sampling_offsets = nn.Linear(d_model, n_heads * n_levels * n_points * 2)
        self.attention_weights = nn.Linear(d_model, n_heads * n_levels * n_points)
        self.value_proj = nn.Linear(d_model, d_model)
        self.output_proj = nn.Linear(d_model, d_model)
        self._reset_parameters()

    def _reset_parameters(self):
        nn.init.constant_(self.sampling_offsets.weight.data, 0.0)
        thetas = torch.arange(self.n_heads, dtype=torch.float32) * (2.0 * math.pi / self.n_heads)
        grid = torch.stack([thetas.cos(), thetas.sin()], -1)
        grid = (grid / grid.abs().max(-1, keepdim=True)[0]).view(self.n_heads, 1, 1, 2).repeat(1, self.n_levels, self.n_points, 1)
        for i in range(self.n_points):
            grid[:, :, i, :] *= i + 1
        with torch.no_grad():
            self.sampling_offsets.bias = nn.Parameter(grid.view(-1))
        nn.init.constant_(self.attention_weights.weight.data, 0.0)
        nn.init.constant_(self.attention_weights.bias.data, 0.0)
        nn.init.xavier_uniform_(self.value_proj.weight.data)
        nn.init.constant_(self.value_proj.bias.data, 0.0)
        nn.init.xavier_uniform_(self.output_proj.weight.data)
        nn.init.constant_(self.output_proj.bias.data, 0.0)

    def forward(self, query, reference_points, input_flatten, input_spatial_shapes, input_level_start_index,
                input_padding_mask=None, residual=None):
        """query (N, Lq, C), reference_points (N|1, Lq, L, 2), input_flatten (N, S, C) -> (N, Lq, C).

        `residual` (fp32, optional) is added in the output projection's epilogue."""
        N, Len_q, _ = query.shape
        N, Len_in, _ = input_flatten.shape
        M, L, P = self.n_heads, self.n_levels, self.n_points
        value = ops.linear(input_flatten, self.value_proj.weight, self.value_proj.bias)          # bf16
        if input_padding_mask is not None:
            value = value.masked_fill(input_padding_mask[..., None], 0.0)
        value = value.view(N, Len_in, M, self.d_model // M)
        off = ops.linear(query, self.sampling_offsets.weight, self.sampling_offsets.bias, out_dtype=torch.float32)
        aw = ops.linear(query, self.attention_weights.weight, self.attention_weights.bias, out_dtype=torch.float32)
        off = off.view(N, Len_q, M, L, P, 2)
        aw = F.softmax(aw.view(N, Len_q, M, L * P), -1).view(N, Len_q, M, L, P)
        if reference_points.shape[-1] == 2:
            normalizer = torch.stack([input_spatial_shapes[..., 1], input_spatial_shapes[..., 0]], -1).to(off.dtype)
            loc = reference_points[:, :, None, :, None, :] + off / normalizer[None, None, None, :, None, :]
        elif reference_points.shape[-1] == 4:
            loc = reference_points[:, :, None, :, None, :2] + off / P * reference_points[:, :, None, :, None, 2:] * 0.5
        else:
            raise ValueError("Last dim of reference_points must be 2 or 4, but get {} instead.".format(reference_points.shape[-1]))
        out = MSDeformAttnFunction.apply(value, input_spatial_shapes, input_level_start_index, loc.contiguous(),
                                         aw.contiguous(), self.im2col_step)
        if residual is not None:
            return ops.linear(out, self.output_proj.weight, self.output_proj.bias, residual=residual)
        return ops.linear(out, self.output_proj.weight, self.output_proj.bias, out_dtype=torch.float32)
