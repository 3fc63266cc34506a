"""The DiNAT oracle (oracle/dinat_ref.py) against itself three ways: NATTEN's arithmetic is not in the reference
(natten==0.14.4, un-vendored, not installed) and the reference has no fixture for it -- PARITY UNPINNED -- so the restatement
is at least checked for internal consistency: gather form == dense masked attention == scalar loops over NATTEN's
get_window_start / get_pb_start formulas, including borders, dilation and ragged residue classes."""
import torch
import pytest

from oracle import dinat_ref as D


def natten_window_start(index, length, K, NS, d):
    """NATTEN 0.14.4 get_window_start, restated literally (scalar)."""
    if d <= 1:
        return max(index - NS, 0) + (index + NS >= length) * (length - index - NS - 1)
    ni = index - NS * d
    if ni < 0:
        return index % d
    if index + NS * d >= length:
        imodd = index % d
        a = (length // d) * d
        b = length - a
        if imodd < b:
            return length - b + imodd - 2 * NS * d
        return a + imodd - K * d
    return ni


def natten_pb_start(index, length, K, NS, d):
    """NATTEN 0.14.4 get_pb_start, restated literally (scalar)."""
    if d <= 1:
        return NS + (index < NS) * (NS - index) + (index + NS >= length) * (length - index - 1 - NS)
    if index - NS * d < 0:
        return K - 1 - (index // d)
    if index + NS * d >= length:
        return (length - index - 1) // d
    return NS


@pytest.mark.parametrize("length,k,d", [(7, 7, 1), (16, 7, 1), (23, 7, 2), (29, 7, 3), (21, 3, 4), (40, 5, 7), (26, 13, 2), (64, 7, 8)])
def test_axis_neighbours_match_natten_formulas(length, k, d):
    nb, pb = D.axis_neighbours(length, k, d)
    for i in range(length):
        ws, ps = natten_window_start(i, length, k, k // 2, d), natten_pb_start(i, length, k, k // 2, d)
        assert nb[i].tolist() == [ws + j * d for j in range(k)], (i, nb[i].tolist(), ws)
        assert pb[i].tolist() == [ps + j for j in range(k)], (i, pb[i].tolist(), ps)
        assert 0 <= min(nb[i]) and max(nb[i]) < length and 0 <= min(pb[i]) and max(pb[i]) < 2 * k - 1


@pytest.mark.parametrize("H,W,k,d", [(9, 11, 3, 1), (8, 13, 3, 2), (7, 7, 7, 1), (15, 14, 7, 2), (15, 17, 5, 3)])
def test_na2d_gather_equals_dense_masked_attention(H, W, k, d):
    g = torch.Generator().manual_seed(H * 100 + W)
    q, kk, v = (torch.randn(2, 2, H, W, 8, generator=g) for _ in range(3))
    rpb = torch.randn(2, 2 * k - 1, 2 * k - 1, generator=g)
    a = D.na2d(q, kk, v, rpb, k, d)
    b = D.na2d_dense(q, kk, v, rpb, k, d)
    torch.testing.assert_close(a, b, atol=1e-5, rtol=1e-5)


def test_na2d_scalar_loops():
    H, W, k, d, hd = 9, 10, 3, 2, 4
    g = torch.Generator().manual_seed(5)
    q, kk, v = (torch.randn(1, 1, H, W, hd, generator=g) for _ in range(3))
    rpb = torch.randn(1, 2 * k - 1, 2 * k - 1, generator=g)
    got = D.na2d(q, kk, v, rpb, k, d)[0, 0]
    for y in range(H):
        for x in range(W):
            sy, sx = natten_window_start(y, H, k, k // 2, d), natten_window_start(x, W, k, k // 2, d)
            by, bx = natten_pb_start(y, H, k, k // 2, d), natten_pb_start(x, W, k, k // 2, d)
            s = torch.tensor([[float(q[0, 0, y, x] @ kk[0, 0, sy + i * d, sx + j * d]) + float(rpb[0, by + i, bx + j]) for j in range(k)]
                              for i in range(k)])
            a = s.flatten().softmax(0).reshape(k, k)
            want = sum(a[i, j] * v[0, 0, sy + i * d, sx + j * d] for i in range(k) for j in range(k))
            torch.testing.assert_close(got[y, x], want, atol=1e-5, rtol=1e-5)


def test_neighborhood_attention_pads_small_inputs_like_natten():
    """Inputs smaller than kernel_size * dilation are zero-padded right / bottom before qkv and cropped after."""
    cfg = D.DiNATCfg(16, 2.0, (1,), (2,), 3, ((2,),), (0,))
    g = torch.Generator().manual_seed(0)
    sd = {k: torch.randn(s, generator=g) * 0.2 for k, s in D.dinat_param_shapes(cfg).items()}
    x = torch.randn(1, 4, 5, 16, generator=g)                       # 4 x 5 < 6 x 6
    y = D.neighborhood_attention(x, sd, "backbone.levels.0.blocks.0.attn", 2, 3, 2)
    assert y.shape == x.shape and torch.isfinite(y).all()
    xp = torch.nn.functional.pad(x, (0, 0, 0, 1, 0, 2))
    yp = D.neighborhood_attention(xp, sd, "backbone.levels.0.blocks.0.attn", 2, 3, 2)
    torch.testing.assert_close(y, yp[:, :4, :5], atol=1e-5, rtol=1e-5)


def test_dinat_backbone_shapes_and_grad():
    cfg = D.DiNATCfg(32, 2.0, (1, 1, 2, 1), (1, 2, 4, 8), 3, ((1,), (2,), (1, 2), (1,)))
    g = torch.Generator().manual_seed(1)
    sd = {k: (torch.randn(s, generator=g) * 0.1).requires_grad_() for k, s in D.dinat_param_shapes(cfg).items()}
    img = torch.randn(1, 3, 128, 192, generator=g)
    outs = D.dinat_backbone(img, sd, cfg)
    assert [tuple(outs[f"res{i}"].shape) for i in (2, 3, 4, 5)] == [(1, 32, 32, 48), (1, 64, 16, 24), (1, 128, 8, 12), (1, 256, 4, 6)]
    sum(o.square().mean() for o in outs.values()).backward()
    assert all(v.grad is not None and torch.isfinite(v.grad).all() for v in sd.values())
