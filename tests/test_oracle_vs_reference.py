"""Live check of the oracle against the reference's own modules (build container only).

Skipped wherever /root/reference is absent (the GPU box).  Complements the fixtures with
shapes the fixtures do not hold: other window sizes, shifts and odd feature maps.
"""
import warnings

import pytest
import torch

from oracle import fill, ref_loader
from oracle import torch_ref as T

pytestmark = pytest.mark.skipif(not ref_loader.available(), reason="reference tree not present")


@pytest.fixture(scope="module")
def ref():
    warnings.filterwarnings("ignore")
    return ref_loader.load()


@pytest.mark.parametrize("C,nH,ws,H,W", [(32, 1, 4, 9, 11), (64, 2, 7, 14, 14), (64, 2, 12, 5, 30), (96, 3, 3, 6, 9)])
def test_basic_layer(ref, C, nH, ws, H, W):
    layer = ref.swin.BasicLayer(dim=C, depth=2, num_heads=nH, window_size=ws, drop_path=0.0)
    layer.eval()
    fill.fill_module(layer, "backbone.layers.0.")
    sd = {"backbone.layers.0." + k: v for k, v in layer.state_dict().items()}
    x = torch.randn(2, H * W, C, generator=torch.Generator().manual_seed(H * W))
    with torch.no_grad():
        want = layer(x, H, W)[0]
        y = T.swin_block(x, sd, "backbone.layers.0.blocks.0", H, W, ws, 0, nH)
        y = T.swin_block(y, sd, "backbone.layers.0.blocks.1", H, W, ws, ws // 2, nH)
    torch.testing.assert_close(y, want, atol=2e-5, rtol=1e-5)


def test_relative_position_index(ref):
    for ws in (3, 7, 12):
        wa = ref.swin.WindowAttention(32, (ws, ws), 1)
        assert (wa.relative_position_index == T.relative_position_index(ws)).all()


def test_swin_with_padding_and_odd_merges(ref):
    cfg = T.SwinCfg(32, (2, 2, 2, 2), (1, 2, 4, 8), 5)
    m = ref.swin.SwinTransformer(embed_dim=32, depths=[2, 2, 2, 2], num_heads=[1, 2, 4, 8], window_size=5)
    m.eval()
    fill.fill_module(m, "backbone.")
    sd = {"backbone." + k: v for k, v in m.state_dict().items()}
    x = torch.randn(1, 3, 70, 107, generator=torch.Generator().manual_seed(1))   # not a multiple of 4
    with torch.no_grad():
        want, got = m(x), T.swin_backbone(x, sd, cfg)
    for k in want:
        torch.testing.assert_close(got[k], want[k], atol=3e-5, rtol=1e-5)


def test_msdeform_core_matches_grid_sample_path(ref):
    g = torch.Generator().manual_seed(3)
    shapes = [(5, 7), (3, 4)]
    S = sum(h * w for h, w in shapes)
    value = torch.randn(2, S, 4, 8, generator=g)
    loc = torch.rand(2, 13, 4, 2, 3, 2, generator=g) * 1.4 - 0.2
    w = torch.rand(2, 13, 4, 2, 3, generator=g)
    want = ref.msda_func.ms_deform_attn_core_pytorch(value, shapes, loc, w)
    torch.testing.assert_close(T.ms_deform_attn_core(value, shapes, loc, w), want, atol=1e-5, rtol=1e-5)
