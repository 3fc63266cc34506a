"""Sine position embedding (reference transformer_decoder/position_encoding.py:15-55).

Input independent: depends only on (H, W), so it is generated once per shape on the device and
cached, instead of being rebuilt by ~10 ATen kernels on every call as in the reference."""
import math

import torch
from torch import nn


class PositionEmbeddingSine(nn.Module):
    def __init__(self, num_pos_feats=64, temperature=10000, normalize=False, scale=None):
        super().__init__()
        if scale is not None and normalize is False:
            raise ValueError("normalize should be True if scale is passed")
        self.num_pos_feats, self.temperature, self.normalize = num_pos_feats, temperature, normalize
        self.scale = 2 * math.pi if scale is None else scale
        self._cache = {}

    def _table(self, H, W, device):
        key = (H, W, str(device))
        t = self._cache.get(key)
        if t is None:
            y = torch.arange(1, H + 1, dtype=torch.float32, device=device)
            x = torch.arange(1, W + 1, dtype=torch.float32, device=device)
            if self.normalize:
                eps = 1e-6
                y = y / (float(H) + eps) * self.scale
                x = x / (float(W) + eps) * self.scale
            d = torch.arange(self.num_pos_feats, dtype=torch.float32, device=device)
            d = self.temperature ** (2 * torch.div(d, 2, rounding_mode="floor") / self.num_pos_feats)
            px, py = x[:, None] / d, y[:, None] / d
            px = torch.stack((px[:, 0::2].sin(), px[:, 1::2].cos()), 2).flatten(1)
            py = torch.stack((py[:, 0::2].sin(), py[:, 1::2].cos()), 2).flatten(1)
            t = torch.cat((py[:, None, :].expand(H, W, -1), px[None, :, :].expand(H, W, -1)), 2).contiguous()  # (H, W, 2F)
            self._cache[key] = t
        return t

    def tokens(self, B, H, W, device):
        """(B, H*W, 2F) channels-last view (no copy across the batch)."""
        return self._table(H, W, device).view(1, H * W, -1).expand(B, -1, -1)

    def forward(self, x, mask=None):
        if mask is not None:
            raise NotImplementedError("padding masks are not used on the segmentation path (oneformer_transformer_decoder.py:413)")
        B, _, H, W = x.shape
        return self._table(H, W, x.device).permute(2, 0, 1)[None].expand(B, -1, -1, -1)
