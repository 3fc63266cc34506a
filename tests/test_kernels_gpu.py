"""HIP kernels vs a plain fp32 PyTorch statement of the same op (GPU box, through the C ABI).

Tolerances: operands are bf16 (8 significand bits, rel. rounding 2^-9) with fp32 accumulation, so a
K-term dot product of O(1) values carries ~2^-9 * sqrt(K) absolute error; outputs stored as bf16
add one more 2^-9 relative rounding.  References are computed in fp32 from the *bf16-rounded*
operands so only accumulation order and the output rounding differ.
"""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def K():
    from uenc import kernels
    return kernels


def _r(*shape, seed=0, scale=1.0, dtype=torch.float32):
    g = torch.Generator(device="cpu").manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(dtype).cuda()


def _close(got, want, atol, rtol):
    torch.testing.assert_close(got.float().cpu(), want.float().cpu(), atol=atol, rtol=rtol)


@pytest.mark.parametrize("M,N,K_", [(128, 128, 64), (256, 384, 192), (200, 100, 72), (1000, 576, 192), (77, 20, 256), (16, 3072, 768)])
@pytest.mark.parametrize("a_dtype", [torch.bfloat16, torch.float32])
def test_gemm_nt_plain(K, M, N, K_, a_dtype):
    a = _r(M, K_, seed=1, dtype=a_dtype)
    w = _r(N, K_, seed=2, scale=K_ ** -0.5, dtype=torch.bfloat16)
    bias = _r(N, seed=3)
    want = a.to(torch.bfloat16).float() @ w.float().t() + bias
    got = K.gemm_nt(a, w, bias=bias, out_dtype=torch.float32)
    _close(got, want, 2e-3, 2e-3)
    got16 = K.gemm_nt(a, w, bias=bias, out_dtype=torch.bfloat16)
    _close(got16, want, 2e-2, 1e-2)


def test_gemm_nt_asymmetric_identity(K):
    # A = I, asymmetric W: catches a transposed fragment / C-write (cdna_hip_programming.md §3)
    n = 128
    a = torch.eye(n, dtype=torch.bfloat16).cuda()
    w = (torch.arange(n * n, dtype=torch.float32).reshape(n, n) % 251 - 125).to(torch.bfloat16).cuda()
    got = K.gemm_nt(a, w, out_dtype=torch.float32)
    _close(got, w.float().t(), 0, 0)


def test_gemm_nt_epilogues(K):
    M, N, K_ = 300, 256, 128
    a = _r(M, K_, seed=1, dtype=torch.bfloat16)
    w = _r(N, K_, seed=2, scale=K_ ** -0.5, dtype=torch.bfloat16)
    bias = _r(N, seed=3)
    pre = a.float() @ w.float().t() + bias
    pre_out = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
    got = K.gemm_nt(a, w, bias=bias, epilogue=K.EPI_GELU, aux_out=pre_out)
    _close(got, torch.nn.functional.gelu(pre), 2e-2, 1e-2)
    _close(pre_out, pre, 2e-2, 1e-2)
    _close(K.gemm_nt(a, w, bias=bias, epilogue=K.EPI_RELU), pre.relu(), 2e-2, 1e-2)
    res = _r(M, N, seed=4)
    _close(K.gemm_nt(a, w, bias=bias, epilogue=K.EPI_RESIDUAL, aux=res, out_dtype=torch.float32), pre + res, 2e-3, 2e-3)
    # in-place residual update (aux aliases out)
    res2 = res.clone()
    K.gemm_nt(a, w, bias=bias, epilogue=K.EPI_RESIDUAL, aux=res2, out=res2)
    _close(res2, pre + res, 2e-3, 2e-3)
    saved = _r(M, N, seed=5, dtype=torch.bfloat16)
    x = saved.float().requires_grad_()
    torch.nn.functional.gelu(x).sum().backward()
    _close(K.gemm_nt(a, w, epilogue=K.EPI_MUL_DGELU, aux=saved), (pre - bias) * x.grad, 3e-2, 2e-2)
    _close(K.gemm_nt(a, w, epilogue=K.EPI_MUL_DRELU, aux=saved), (pre - bias) * (saved.float() > 0), 3e-2, 2e-2)


def test_gemm_nt_splitk_and_strided(K):
    M, N, K_ = 150, 256, 4096
    a = _r(M, K_, seed=1, dtype=torch.bfloat16)
    w = _r(N, K_, seed=2, scale=K_ ** -0.5, dtype=torch.bfloat16)
    want = a.float() @ w.float().t()
    _close(K.gemm_nt(a, w, out_dtype=torch.float32, splitk=8), want, 3e-3, 3e-3)
    # strided A (a slice of a packed qkv buffer) and strided output
    big = _r(M, 3 * 256, seed=6, dtype=torch.bfloat16)
    w2 = _r(64, 256, seed=7, scale=1 / 16, dtype=torch.bfloat16)
    out = torch.zeros(M, 192, dtype=torch.bfloat16, device="cuda")
    K.gemm_nt(big[:, 256:512], w2, out=out[:, 64:128])
    _close(out[:, 64:128], big[:, 256:512].float() @ w2.float().t(), 2e-2, 1e-2)
    assert float(out[:, :64].abs().max()) == 0 and float(out[:, 128:].abs().max()) == 0


@pytest.mark.parametrize("M,N,K_", [(128, 128, 128), (1000, 192, 576), (4096, 264, 96), (130, 24, 1032)])
@pytest.mark.parametrize("x_dtype", [torch.bfloat16, torch.float32])
def test_gemm_tn_wgrad(K, M, N, K_, x_dtype):
    dy = _r(M, N, seed=1, dtype=torch.bfloat16)
    x = _r(M, K_, seed=2, dtype=x_dtype)
    dw = _r(N, K_, seed=3)
    db = _r(N, seed=4)
    want_w = dw + dy.float().t() @ x.to(torch.bfloat16).float()
    want_b = db + dy.float().sum(0)
    K.gemm_tn(dy, x, dw, db)
    tol = 4e-3 * (M ** 0.5)
    _close(dw, want_w, tol, 2e-3)
    _close(db, want_b, tol, 2e-3)


def test_gemm_tn_asymmetric(K):
    # dy = I (M = N): dW must equal X exactly
    n = 128
    dy = torch.eye(n, dtype=torch.bfloat16).cuda()
    x = (torch.arange(n * 64, dtype=torch.float32).reshape(n, 64) % 253 - 126).to(torch.bfloat16).cuda()
    dw = torch.zeros(n, 64, device="cuda")
    K.gemm_tn(dy, x, dw, None)
    _close(dw, x.float(), 0, 0)


@pytest.mark.parametrize("M,C", [(1000, 96), (333, 192), (64, 256), (50, 768), (20, 1536), (9, 3072), (5, 6144)])
def test_layernorm_fwd_bwd(K, M, C):
    x = _r(M, C, seed=1, scale=2.0) + 0.5
    res = _r(M, C, seed=2)
    gamma = 1 + 0.1 * _r(C, seed=3)
    beta = 0.1 * _r(C, seed=4)
    xr = (x + res).requires_grad_()
    want = torch.nn.functional.layer_norm(xr, (C,), gamma, beta, 1e-5)
    y, h, stats = K.layernorm_fwd(x, gamma, beta, res=res, out_dtype=torch.float32, want_h=True)
    _close(h, xr, 1e-6, 1e-6)
    _close(y, want, 2e-5, 2e-5)
    y16, _, _ = K.layernorm_fwd(x, gamma, beta, res=res, out_dtype=torch.bfloat16)
    _close(y16, want, 2e-2, 1e-2)
    dy = _r(M, C, seed=5)
    gam = gamma.clone().requires_grad_()
    bet = beta.clone().requires_grad_()
    torch.nn.functional.layer_norm(xr, (C,), gam, bet, 1e-5).backward(dy)
    dres = _r(M, C, seed=6)
    dg, db = torch.zeros(C, device="cuda"), torch.zeros(C, device="cuda")
    dx = K.layernorm_bwd(dy, h, stats, gamma, dres=dres, dgamma=dg, dbeta=db)
    _close(dx, xr.grad + dres, 2e-4, 1e-4)
    _close(dg, gam.grad, 1e-3, 1e-4)
    _close(db, bet.grad, 1e-3, 1e-4)
    # bf16 gradient input
    dx16 = K.layernorm_bwd(dy.to(torch.bfloat16), h, stats, gamma)
    _close(dx16, xr.grad, 3e-2, 2e-2)


def test_casts(K):
    w = _r(96, 200, seed=1)
    _close(K.cast_bf16(w), w.to(torch.bfloat16), 0, 0)
    _close(K.cast_transpose_bf16(w), w.t().to(torch.bfloat16), 0, 0)
    b = _r(20, seed=2)
    _close(K.cast_bf16(b), b.to(torch.bfloat16), 0, 0)


def _window_attn_ref(qkv, qkv_bias, table, H, W, ws, shift, nH):
    """fp32 statement of the attention core on (B, H*W, 3C) qkv, padding slots = bias; uses the oracle's layout."""
    from oracle import torch_ref as T
    B, L, C3 = qkv.shape
    C, hd = C3 // 3, C3 // 3 // nH
    src, rid = T.window_layout(H, W, ws, shift)
    src, rid = src.to(qkv.device), rid.to(qkv.device)
    nW, N = src.shape
    z = torch.cat([qkv, qkv_bias.view(1, 1, C3).expand(B, 1, C3)], 1)
    w = z[:, src.reshape(-1)].view(B * nW, N, 3, nH, hd).permute(2, 0, 3, 1, 4)
    q, k, v = w[0], w[1], w[2]
    attn = (q @ k.transpose(-1, -2)) * hd ** -0.5
    bias = table[T.relative_position_index(ws).reshape(-1).to(qkv.device)].view(N, N, nH).permute(2, 0, 1)
    attn = attn + bias[None]
    if shift > 0:
        m = (rid[:, :, None] != rid[:, None, :]).float() * -100.0
        attn = (attn.view(B, nW, nH, N, N) + m[None, :, None]).view(B * nW, nH, N, N)
    o = (attn.softmax(-1) @ v).transpose(1, 2).reshape(B, nW * N, C)
    y = o.new_zeros(B, L + 1, C)
    y[:, src.reshape(-1)] = o
    return y[:, :L]


# the last two geometries give every persistent backward workgroup SEVERAL windows (160 windows on 42 / 21 groups per head): stage
# alternation, the loader wave's DMA of the next window and its three rotating slot tables (csrc/window_attn.hip) are only exercised then
@pytest.mark.parametrize("ws,H,W,nH,shift", [(7, 24, 40, 3, 0), (7, 24, 40, 3, 3), (12, 20, 30, 2, 0), (12, 20, 30, 2, 6),
                                              (12, 24, 36, 1, 6), (3, 7, 8, 1, 1), (5, 9, 11, 2, 2), (8, 16, 16, 1, 4),
                                              (12, 96, 120, 6, 6), (12, 91, 118, 12, 0)])
def test_window_attention_fwd_bwd(K, ws, H, W, nH, shift):
    B, C = 2, nH * 32
    qkv = _r(B, H * W, 3 * C, seed=1, scale=1.5, dtype=torch.bfloat16)
    qb = _r(3 * C, seed=2, scale=0.5, dtype=torch.bfloat16)
    table = _r((2 * ws - 1) ** 2, nH, seed=3, scale=0.5)
    bq, bk = K.relpos_expand(table, ws)
    out = K.window_attn_fwd(qkv.view(B, H, W, 3 * C), qb, bq, ws, shift, 32 ** -0.5)
    q32 = qkv.float().requires_grad_()
    b32 = qb.float().requires_grad_()
    t32 = table.clone().requires_grad_()
    want = _window_attn_ref(q32, b32, t32, H, W, ws, shift, nH)
    _close(out.view(B, H * W, C), want, 3e-2, 2e-2)
    do = _r(B, H * W, C, seed=4, dtype=torch.bfloat16)
    want.backward(do.float())
    # the two parameter gradients are ACCUMULATED into the caller's buffers (the parameters' .grad): start them non-zero
    dtab0, dpad0 = _r((2 * ws - 1) ** 2, nH, seed=5), _r(3 * C, seed=6)
    dtab, dpad = dtab0.clone(), dpad0.clone()
    dqkv = K.window_attn_bwd(qkv.view(B, H, W, 3 * C), qb, bq, bk, out, do.view(B, H, W, C), ws, shift, 32 ** -0.5, dtable=dtab, dbias=dpad)
    gs = float(q32.grad.abs().max())
    _close(dqkv.view(B, H * W, 3 * C), q32.grad, 3e-2 * gs, 3e-2)
    _close(dtab - dtab0, t32.grad, 3e-2 * float(t32.grad.abs().max()) + 1e-3, 3e-2)
    _close(dpad - dpad0, b32.grad, 3e-2 * float(b32.grad.abs().max()) + 1e-3, 3e-2)
    # the same with the forward's saved row statistics (12 x 12 windows: the backward skips its maximum / sum passes; others ignore them)
    out2, lse = K.window_attn_fwd(qkv.view(B, H, W, 3 * C), qb, bq, ws, shift, 32 ** -0.5, want_lse=True)
    assert torch.equal(out2, out) and lse.shape == (B, H, W, nH) and bool(torch.isfinite(lse).all())
    dtab, dpad = dtab0.clone(), dpad0.clone()
    dqkv = K.window_attn_bwd(qkv.view(B, H, W, 3 * C), qb, bq, bk, out, do.view(B, H, W, C), ws, shift, 32 ** -0.5, dtable=dtab, dbias=dpad, lse=lse)
    _close(dqkv.view(B, H * W, 3 * C), q32.grad, 3e-2 * gs, 3e-2)
    _close(dtab - dtab0, t32.grad, 3e-2 * float(t32.grad.abs().max()) + 1e-3, 3e-2)
    _close(dpad - dpad0, b32.grad, 3e-2 * float(b32.grad.abs().max()) + 1e-3, 3e-2)


@pytest.mark.parametrize("H,W,nH,shift", [(96, 120, 6, 6), (40, 50, 24, 0)])
def test_window_attention_bwd_kernel_forms_agree(K, H, W, nH, shift, monkeypatch):
    """12 x 12 windows: the loader-wave kernel (production), the nine-wave kernel with the LDS bias table (variant 4) and the one with
    dense bias rows (variant 2) compute the same arithmetic in the same order: dqkv bit-identical, the two parameter gradients equal
    up to the order of their float atomics."""
    B, C, ws = 2, nH * 32, 12
    qkv = _r(B, H * W, 3 * C, seed=11, scale=1.5, dtype=torch.bfloat16)
    qb = _r(3 * C, seed=12, scale=0.5, dtype=torch.bfloat16)
    table = _r((2 * ws - 1) ** 2, nH, seed=13, scale=0.5)
    bq, bk = K.relpos_expand(table, ws)
    do = _r(B, H * W, C, seed=14, dtype=torch.bfloat16)
    got = {}
    for variant in ("0", "4", "2"):
        monkeypatch.setenv("UENC_WATTN_VARIANT", variant)
        out = K.window_attn_fwd(qkv.view(B, H, W, 3 * C), qb, bq, ws, shift, 32 ** -0.5)
        dtab, dpad = torch.zeros((2 * ws - 1) ** 2, nH, device="cuda"), torch.zeros(3 * C, device="cuda")
        dqkv = K.window_attn_bwd(qkv.view(B, H, W, 3 * C), qb, bq, bk, out, do.view(B, H, W, C), ws, shift, 32 ** -0.5, dtable=dtab, dbias=dpad)
        torch.cuda.synchronize()
        got[variant] = (out.clone(), dqkv.clone(), dtab, dpad)
    monkeypatch.delenv("UENC_WATTN_VARIANT")
    for variant in ("4", "2"):
        assert torch.equal(got["0"][0], got[variant][0]), f"forward differs from variant {variant}"
        assert torch.equal(got["0"][1], got[variant][1]), f"dqkv differs from variant {variant}"
        _close(got["0"][2], got[variant][2], 1e-4 * float(got["0"][2].abs().max()) + 1e-6, 1e-4)
        _close(got["0"][3], got[variant][3], 1e-4 * float(got["0"][3].abs().max()) + 1e-6, 1e-4)


def test_msdeform_golden(K):
    """The reference's own forward + autograd gradients (tests/golden/msdeform_core.npz)."""
    from conftest import load_golden
    g = load_golden("msdeform_core")
    shapes = g["shapes"].cuda()
    start = torch.cat([shapes.new_zeros(1), (shapes[:, 0] * shapes[:, 1]).cumsum(0)[:-1]])
    value, loc, w, go = (g[k].cuda().contiguous() for k in ("value", "loc", "w", "grad_out"))
    out = K.msdeform_attn_fwd(value, shapes, start, loc, w)
    _close(out, g["out"], 1e-5, 1e-5)
    gv, gl, ga = K.msdeform_attn_bwd(value, shapes, start, loc, w, go)
    _close(gv, g["grad_value"], 1e-5, 1e-4)
    _close(ga, g["grad_w"], 1e-5, 1e-4)
    _close(gl, g["grad_loc"], 2e-4, 1e-4)
    # bf16 value / bf16 output variant
    out16 = K.msdeform_attn_fwd(value.to(torch.bfloat16), shapes, start, loc, w, out_dtype=torch.bfloat16)
    _close(out16, g["out"], 3e-2, 2e-2)


def test_msdeform_vs_oracle_random(K):
    from oracle import torch_ref as T
    shapes_l = [(5, 9), (10, 18), (20, 36)]
    S = sum(h * w for h, w in shapes_l)
    B, Lq, M, D, L, P = 2, S, 8, 32, 3, 4
    gen = torch.Generator().manual_seed(0)
    value = torch.randn(B, S, M, D, generator=gen)
    loc = torch.rand(B, Lq, M, L, P, 2, generator=gen) * 1.2 - 0.1
    w = torch.rand(B, Lq, M, L, P, generator=gen).view(B, Lq, M, -1).softmax(-1).view(B, Lq, M, L, P)
    v32, l32, w32 = value.clone().requires_grad_(), loc.clone().requires_grad_(), w.clone().requires_grad_()
    want = T.ms_deform_attn_core(v32, shapes_l, l32, w32)
    go = torch.randn(want.shape, generator=gen)
    want.backward(go)
    shapes = torch.tensor(shapes_l, dtype=torch.int64).cuda()
    start = torch.cat([shapes.new_zeros(1), (shapes[:, 0] * shapes[:, 1]).cumsum(0)[:-1]])
    out = K.msdeform_attn_fwd(value.cuda(), shapes, start, loc.cuda(), w.cuda())
    _close(out, want, 1e-5, 1e-5)
    gv, gl, ga = K.msdeform_attn_bwd(value.cuda(), shapes, start, loc.cuda(), w.cuda(), go.cuda())
    _close(gv, v32.grad, 1e-4, 1e-4)
    _close(ga, w32.grad, 1e-5, 1e-4)
    _close(gl, l32.grad, 5e-4, 1e-4)


@pytest.mark.parametrize("shapes_l,spread", [([(5, 9), (10, 18), (20, 36)], 3.0), ([(8, 16), (16, 32), (32, 64)], 4.0),
                                             ([(7, 11), (13, 22), (27, 43)], 40.0), ([(24, 40)], 2.0)])
def test_msdeform_bwd_tiled_vs_oracle(K, shapes_l, spread):
    """Encoder case (queries = pixels of the pyramid, samples a few pixels around the query): the LDS-tiled backward,
    with in-window and out-of-window samples, against the oracle's autograd and against the direct-atomics kernel."""
    from oracle import torch_ref as T
    L = len(shapes_l)
    S = sum(h * w for h, w in shapes_l)
    B, M, D, P = 2, 8, 32, 4
    gen = torch.Generator().manual_seed(1)
    ref = torch.cat([torch.stack(torch.meshgrid((torch.arange(h) + 0.5) / h, (torch.arange(w) + 0.5) / w, indexing="ij"), -1)
                     .reshape(-1, 2).flip(-1) for h, w in shapes_l])                          # (S, 2) as (x, y)
    norm = torch.tensor([[w, h] for h, w in shapes_l], dtype=torch.float32)                     # (L, 2)
    off = (torch.rand(B, S, M, L, P, 2, generator=gen) * 2 - 1) * spread
    off[:, :, :, :, 0] = off[:, :, :, :, 0].round()                                            # integer offsets: lh = lw = 0 taps
    loc = (ref[None, :, None, None, None, :] + off / norm[None, None, None, :, None, :]).contiguous()
    value = torch.randn(B, S, M, D, generator=gen)
    w = torch.rand(B, S, M, L * P, generator=gen).softmax(-1).view(B, S, M, L, P)
    v32, l32, w32 = value.clone().requires_grad_(), loc.clone().requires_grad_(), w.clone().requires_grad_()
    want = T.ms_deform_attn_core(v32, shapes_l, l32, w32)
    go = torch.randn(want.shape, generator=gen)
    want.backward(go)
    shapes = torch.tensor(shapes_l, dtype=torch.int64).cuda()
    start = torch.cat([shapes.new_zeros(1), (shapes[:, 0] * shapes[:, 1]).cumsum(0)[:-1]])
    args = (value.cuda(), shapes, start, loc.cuda(), w.cuda(), go.cuda())
    gv, gl, ga = K.msdeform_attn_bwd(*args, shapes_host=shapes_l)
    scale = float(v32.grad.abs().max())
    _close(gv, v32.grad, 2e-5 * scale + 1e-5, 1e-4)
    _close(ga, w32.grad, 1e-4, 1e-4)
    # (point 0 sits exactly on pixel centres, where d/dloc of the bilinear kernel is one-sided: compared below instead)
    _close(gl[:, :, :, :, 1:], l32.grad[:, :, :, :, 1:], 5e-4 * float(l32.grad.abs().max()), 1e-4)
    gv0, gl0, ga0 = K.msdeform_attn_bwd(*args)
    _close(gv, gv0, 2e-5 * scale + 1e-5, 1e-4)
    _close(gl, gl0, 1e-4 * float(gl0.abs().max()), 1e-4)
    _close(ga, ga0, 1e-4, 1e-4)
    # bf16 value / bf16 grad_out operands (what the pixel decoder passes)
    a16 = (args[0].to(torch.bfloat16), shapes, start, args[3], args[4], args[5].to(torch.bfloat16))
    gv1, gl1, ga1 = K.msdeform_attn_bwd(*a16, shapes_host=shapes_l)
    gv2, gl2, ga2 = K.msdeform_attn_bwd(*a16)
    _close(gv1, gv2, 2e-5 * scale + 1e-5, 1e-4)
    _close(gl1, gl2, 1e-4 * float(gl2.abs().max()), 1e-4)
    _close(ga1, ga2, 1e-4, 1e-4)


def test_gemm_tn_f32_dy(K):
    dy = _r(300, 64, seed=1)
    x = _r(300, 40, seed=2)
    dw = torch.zeros(64, 40, device="cuda")
    K.gemm_tn(dy, x, dw, None)
    _close(dw, dy.to(torch.bfloat16).float().t() @ x.to(torch.bfloat16).float(), 5e-2, 2e-3)


@pytest.mark.parametrize("M,N,K_", [(4096, 2560, 128), (5000, 2312, 192), (16384, 768, 768), (3000, 3584, 64 * 5), (41000, 192, 192), (40008, 256, 288)])
def test_gemm_nt_large_tile_path(K, M, N, K_):
    """Shapes that take the 256x256 LDS-DMA kernel (bf16 A, K % 64 == 0, >= 160 tiles), incl. ragged M / N and epilogues."""
    a = _r(M, K_, seed=1, dtype=torch.bfloat16)
    w = _r(N, K_, seed=2, scale=K_ ** -0.5, dtype=torch.bfloat16)
    bias = _r(N, seed=3)
    pre = a.float() @ w.float().t() + bias
    _close(K.gemm_nt(a, w, bias=bias, out_dtype=torch.float32), pre, 2e-3, 2e-3)
    _close(K.gemm_nt(a, w, bias=bias), pre, 2e-2, 1e-2)
    pre_out = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
    _close(K.gemm_nt(a, w, bias=bias, epilogue=K.EPI_GELU, aux_out=pre_out), torch.nn.functional.gelu(pre), 2e-2, 1e-2)
    _close(pre_out, pre, 2e-2, 1e-2)
    res = _r(M, N, seed=4)
    res2 = res.clone()
    K.gemm_nt(a, w, bias=bias, epilogue=K.EPI_RESIDUAL, aux=res2, out=res2)
    _close(res2, pre + res, 2e-3, 2e-3)
    saved = _r(M, N, seed=5, dtype=torch.bfloat16)
    _close(K.gemm_nt(a, w, epilogue=K.EPI_MUL_DRELU, aux=saved), (pre - bias) * (saved.float() > 0), 3e-2, 2e-2)
    # exact check with an asymmetric integer operand (catches fragment / swizzle mistakes)
    ai = (torch.arange(M * K_, dtype=torch.float32).reshape(M, K_) % 7 - 3).to(torch.bfloat16).cuda()
    wi = (torch.arange(N * K_, dtype=torch.float32).reshape(N, K_) % 5 - 2).to(torch.bfloat16).cuda()
    _close(K.gemm_nt(ai, wi, out_dtype=torch.float32), ai.float() @ wi.float().t(), 0, 0)


@pytest.mark.parametrize("M,N,K_", [(40008, 520, 128), (41000, 776, 320), (45000, 200, 192), (16384, 768, 768), (66000, 384, 64 * 7)])
def test_gemm_nt_192_wide_tile(K, M, N, K_, monkeypatch):
    """The 256 x 192 tile of the persistent LDS-DMA kernel (three 16-MFMA phases per k-tile, an unpaired third column group per wave) forced on
    ragged shapes -- last column tile 136 / 8 / 192 wide, 2 ... 12 k-tiles, persistent and one-tile-per-workgroup grids -- against the
    256-wide tile (bit-identical: same k order) and against fp32 matmul, every epilogue."""
    a = _r(M, K_, seed=1, dtype=torch.bfloat16)
    w = _r(N, K_, seed=2, scale=K_ ** -0.5, dtype=torch.bfloat16)
    bias = _r(N, seed=3)
    res = _r(M, N, seed=4)
    saved = _r(M, N, seed=5, dtype=torch.bfloat16)
    pre = a.float() @ w.float().t() + bias

    def run_all():
        outs = [K.gemm_nt(a, w, bias=bias, out_dtype=torch.float32), K.gemm_nt(a, w, bias=bias)]
        po = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
        outs += [K.gemm_nt(a, w, bias=bias, epilogue=K.EPI_GELU, aux_out=po), po]
        outs.append(K.gemm_nt(a, w, bias=bias, epilogue=K.EPI_RELU))
        outs.append(K.gemm_nt(a, w, bias=bias, epilogue=K.EPI_RESIDUAL, aux=res, out_dtype=torch.float32))
        outs.append(K.gemm_nt(a, w, bias=bias, epilogue=K.EPI_RESIDUAL, aux=res, out_dtype=torch.bfloat16))
        outs.append(K.gemm_nt(a, w, epilogue=K.EPI_MUL_DGELU, aux=saved))
        outs.append(K.gemm_nt(a, w, epilogue=K.EPI_MUL_DRELU, aux=saved))
        outs.append(K.gemm_nt(a, w, bias=bias, epilogue=K.EPI_RELU, out_dtype=torch.float32))
        torch.cuda.synchronize()
        return outs

    monkeypatch.setenv("UENC_GEMM_VARIANT", str(262144 | 8388608))
    wide = run_all()
    monkeypatch.setenv("UENC_GEMM_VARIANT", str(262144 | 4194304))
    narrow = run_all()
    for x, y in zip(wide, narrow):
        assert torch.equal(x, y)
    _close(narrow[0], pre, 2e-3, 2e-3)
    _close(narrow[1], pre, 2e-2, 1e-2)
    _close(narrow[2], torch.nn.functional.gelu(pre), 2e-2, 1e-2)
    _close(narrow[3], pre, 2e-2, 1e-2)
    _close(narrow[5], pre + res, 2e-3, 2e-3)
    _close(narrow[6], pre + res, 3e-2, 1e-2)
    _close(narrow[8], (pre - bias) * (saved.float() > 0), 3e-2, 2e-2)
    ai = (torch.arange(M * K_, dtype=torch.float32).reshape(M, K_) % 7 - 3).to(torch.bfloat16).cuda()
    wi = (torch.arange(N * K_, dtype=torch.float32).reshape(N, K_) % 5 - 2).to(torch.bfloat16).cuda()
    _close(K.gemm_nt(ai, wi, out_dtype=torch.float32), ai.float() @ wi.float().t(), 0, 0)
    _close(K.gemm_nt(ai, wi).float(), (ai.float() @ wi.float().t()).to(torch.bfloat16).float(), 0, 0)


@pytest.mark.parametrize("M,N,K_,split", [(150, 256, 32768, 64), (600, 512, 8192, 16), (1000, 256, 4096 + 64, 16)])
def test_gemm_nt_splitk_large_tile(K, M, N, K_, split):
    """Split-K on the 256x256 LDS-DMA kernel (skinny outputs with a long contraction: the mask-embedding gradient)."""
    a = _r(M, K_, seed=1, dtype=torch.bfloat16)
    w = _r(N, K_, seed=2, scale=K_ ** -0.5, dtype=torch.bfloat16)
    bias = _r(N, seed=3)
    want = a.float() @ w.float().t() + bias
    got = K.gemm_nt(a, w, bias=bias, out_dtype=torch.float32, splitk=split)
    _close(got, want, 2e-3 * float(want.abs().max()), 2e-3)
    acc = _r(M, N, seed=4)
    acc0 = acc.clone()
    K.gemm_nt(a, w, out=acc, splitk=split, accumulate=True)
    _close(acc, acc0 + want - bias, 2e-3 * float(want.abs().max()), 2e-3)


@pytest.mark.parametrize("variant", ["0", "8"])
@pytest.mark.parametrize("M,N,K_", [(4096, 512, 256), (2048, 576, 192), (8192, 768, 3072), (6400, 264, 200), (2048, 96, 48)])
def test_gemm_tn_large_tile_path(K, M, N, K_, variant, monkeypatch):
    """wgrad shapes through the LDS-DMA + transposed-read kernels: production dispatch (128x128 tiles for small
    outputs, 256x256 from 20 tiles up) and the 256x256 kernel forced; ragged N / K tiles and db."""
    monkeypatch.setenv("UENC_GEMM_VARIANT", variant)
    dy = _r(M, N, seed=1, dtype=torch.bfloat16)
    x = _r(M, K_, seed=2, dtype=torch.bfloat16)
    dw = _r(N, K_, seed=3)
    db = _r(N, seed=4)
    want_w = dw + dy.float().t() @ x.float()
    want_b = db + dy.float().sum(0)
    K.gemm_tn(dy, x, dw, db)
    tol = 4e-3 * (M ** 0.5)
    _close(dw, want_w, tol, 2e-3)
    _close(db, want_b, tol, 2e-3)
    # exact: small integers, asymmetric
    dyi = (torch.arange(M * N, dtype=torch.float32).reshape(M, N) % 5 - 2).to(torch.bfloat16).cuda()
    xi = (torch.arange(M * K_, dtype=torch.float32).reshape(M, K_) % 3 - 1).to(torch.bfloat16).cuda()
    dwi = torch.zeros(N, K_, device="cuda")
    K.gemm_tn(dyi, xi, dwi, None, splitm=1)
    _close(dwi, dyi.float().t() @ xi.float(), 0, 0)


@pytest.mark.parametrize("shape,size", [((2, 5, 16, 24), (64, 96)), ((1, 3, 7, 9), (28, 36)), ((1, 2, 10, 6), (17, 20))])
def test_upsample_bilinear(K, shape, size):
    x = _r(*shape, seed=1)
    want = torch.nn.functional.interpolate(x, size=size, mode="bilinear", align_corners=False)
    _close(K.upsample_bilinear(x, size), want, 1e-5, 1e-5)


def test_attn_mask_matches_torch(K):
    """resize + threshold + all-blocked-row fix == the torch expression of the reference (:497-505, :454)."""
    g = torch.Generator().manual_seed(5)
    x = torch.randn(2, 7, 64, 96, generator=g).cuda()
    x[0, 2] = -x[0, 2].abs() - 0.1                      # a row that is blocked everywhere
    x[1, 0, :, :48] = x[1, 0, :, :48].abs() + 0.1
    for size in ((32, 48), (16, 24), (8, 12), (64, 96), (2, 3), (9, 13)):
        am = torch.nn.functional.interpolate(x, size=size, mode="bilinear", align_corners=False)
        ref = am.sigmoid().flatten(2) < 0.5
        ref = ref & ~ref.all(-1, keepdim=True)
        got = K.attn_mask(x, size)
        assert got.dtype == torch.bool and got.shape == ref.shape
        diff = got != ref
        # a different rounding of the 4-tap sum may flip the sign only where the resized logit is ~0
        assert (am.flatten(2)[diff].abs() < 1e-6).all() and diff.float().mean() < 1e-4
        assert not got[0, 2].any()


@pytest.mark.parametrize("L,P", [(3, 4), (2, 3)])
def test_msda_prep_kernels(K, L, P):
    """softmax + sampling-location arithmetic of MSDeformAttn.forward (ms_deform_attn.py:98-106) and its adjoint."""
    B, Lq, M = 2, 37, 8
    LP = L * P
    ld = 3 * M * LP
    g = torch.Generator().manual_seed(2)
    offaw = torch.randn(B * Lq, ld, generator=g).cuda().requires_grad_()
    ref = torch.rand(1, Lq, L, 2, generator=g).cuda()
    shapes = torch.tensor([(5 + 3 * l, 7 + 5 * l) for l in range(L)], dtype=torch.int64).cuda()
    loc, aw = K.msda_prep_fwd(offaw.detach(), ref, shapes, B, Lq, M, L, P)
    off = offaw[:, :2 * M * LP].view(B, Lq, M, L, P, 2)
    norm = torch.stack([shapes[:, 1], shapes[:, 0]], -1).float()
    loc2 = ref[:, :, None, :, None, :] + off / norm[None, None, None, :, None, :]
    aw2 = offaw[:, 2 * M * LP:].view(B, Lq, M, LP).softmax(-1).view(B, Lq, M, L, P)
    _close(loc, loc2, 1e-6, 1e-5); _close(aw, aw2, 1e-6, 1e-5)
    dloc, daw = torch.randn(loc.shape, generator=g).cuda(), torch.randn(aw.shape, generator=g).cuda()
    ((loc2 * dloc).sum() + (aw2 * daw).sum()).backward()
    got = K.msda_prep_bwd(dloc, daw, aw, shapes, ld)
    _close(got, offaw.grad, 2e-2 * float(offaw.grad.abs().max()), 1e-2)
    start = torch.tensor([0, 11, 30][:L] if L == 3 else [0, 20], dtype=torch.int64).cuda()
    sums = K.segment_colsum(got, start, Lq, B)
    want = torch.stack([got.float().view(B, Lq, ld)[:, int(start[s]):(int(start[s + 1]) if s + 1 < len(start) else Lq)].sum((0, 1))
                        for s in range(len(start))])
    _close(sums, want, 1e-3 * float(want.abs().max()), 1e-3)


@pytest.mark.parametrize("M,N,Kd,split", [(512, 256, 8192, 32), (150, 256, 4096, 7), (1536, 256, 16384, 43), (300, 64, 1024, 5)])
def test_gemm_nt_splitk_stored_partials(K, M, N, Kd, split):
    """Split-K with stored partial tiles (both tile classes) against the fp32 product of the same bf16 operands; rows at a
    non-power-of-two stride like the stacked gradient matrix of ops.MaskHeadsFn."""
    a = _r(M, Kd + 64, seed=1, scale=0.5, dtype=torch.bfloat16)[:, :Kd]
    w = _r(N, Kd + 64, seed=2, scale=0.5, dtype=torch.bfloat16)[:, :Kd]
    got = K.gemm_nt_splitk(a, w, split)
    want = a.float() @ w.float().t()
    assert float((got - want).norm() / want.norm()) < 1e-3       # fp32 accumulation of exact bf16 products: order only


@pytest.mark.parametrize("M,N,K_,nsamp", [(300, 192, 64, 3), (2048, 256, 128, 2), (49152, 768, 256, 3), (16384, 3072, 256, 2)])
def test_gemm_nt_per_sample_scale(K, M, N, K_, nsamp):
    """uenc_gemm_nt_scaled: alpha multiplied per sample (stochastic depth in the epilogue), on the small, the 128-tile and the 256-tile
    kernels (stored fp32 residual form through the LDS-staged epilogue, bf16 activation-derivative form through the direct one)."""
    L = M // nsamp
    a = _r(M, K_, seed=1, dtype=torch.bfloat16)
    w = _r(N, K_, seed=2, scale=K_ ** -0.5, dtype=torch.bfloat16)
    bias, res = _r(N, seed=3), _r(M, N, seed=4)
    sc = torch.tensor([1.25, 0.0, 2.0][:nsamp], device="cuda")
    rows = sc.repeat_interleave(L)[:, None]
    lin = a.float() @ w.float().t() + bias
    got = K.gemm_nt(a, w, bias=bias, epilogue=K.EPI_RESIDUAL, aux=res, out_dtype=torch.float32, sample_scale=sc, rows_per_sample=L)
    _close(got, res + rows * lin, 3e-3, 2e-3)
    assert torch.equal(got[L:2 * L], res[L:2 * L])                 # the dropped sample's rows are the residual, bit for bit
    pre = _r(M, N, seed=5, dtype=torch.bfloat16)
    got2 = K.gemm_nt(a, w, epilogue=K.EPI_MUL_DRELU, aux=pre, sample_scale=sc, rows_per_sample=L)
    _close(got2, rows * (a.float() @ w.float().t()) * (pre.float() > 0), 3e-2, 2e-2)
    got3 = K.gemm_nt(a, w, bias=bias, alpha=0.5, sample_scale=sc, rows_per_sample=L, out_dtype=torch.float32)
    _close(got3, 0.5 * rows * lin, 3e-3, 2e-3)


@pytest.mark.parametrize("M,N,K_", [(4096, 256, 256), (16384, 768, 256), (320, 64, 72)])
def test_weight_gradient_alpha(K, M, N, K_):
    """alpha of the weight-gradient GEMMs (uenc_gemm_tn_scaled and the `alpha` field of the grouped descriptors): the deferred group,
    the small group and the direct launch."""
    from uenc import ops
    dy = _r(M, N, seed=1, dtype=torch.bfloat16)
    x = _r(M, K_, seed=2, dtype=torch.bfloat16)
    want_w = 1.5 * (dy.float().t() @ x.float())
    want_b = 1.5 * dy.float().sum(0)
    gw, gb = torch.zeros(N, K_, device="cuda"), torch.zeros(N, device="cuda")
    K.gemm_tn(dy, x, gw, gb, alpha=1.5)
    _close(gw, want_w, 2e-3 * float(want_w.abs().max()), 2e-3); _close(gb, want_b, 2e-3 * float(want_b.abs().max()), 2e-3)
    gw2, gb2 = torch.zeros(N, K_, device="cuda"), torch.zeros(N, device="cuda")
    ops._tn(dy, x, gw2, gb2, alpha=1.5)
    ops._tn(dy[: M // 2], x[: M // 2], gw2, gb2, alpha=-1.5)       # a row range, negative alpha: the first half cancels
    ops.flush_wgrads()
    want2 = 1.5 * (dy[M // 2:].float().t() @ x[M // 2:].float())
    _close(gw2, want2, 3e-3 * float(want2.abs().max()), 3e-3)
    _close(gb2, 1.5 * dy[M // 2:].float().sum(0), 3e-3 * float(want_b.abs().max()), 3e-3)


@pytest.mark.parametrize("shapes_l,spread,ref_batch", [([(8, 16), (16, 32), (32, 64)], 4.0, False), ([(7, 11), (13, 22), (27, 43)], 30.0, True)])
def test_msdeform_fused_equals_prep_plus_core(K, shapes_l, spread, ref_batch):
    """The fused entry points (sampling locations and softmaxed weights derived INSIDE the attention kernels from the projection row,
    reference ops/modules/ms_deform_attn.py:99-125) against the pair they replace -- uenc_msda_prep_* + uenc_msdeform_attn_* -- on the
    same inputs: identical forward output, grad_value within the float-atomics order, d(offaw) within bf16 rounding of the last step
    (the fused backward skips the fp32 grad_loc / grad_attn round trip, nothing else); incl. borders, out-of-range samples, a ragged last
    workgroup, reference points with and without a batch dimension, and a padded row stride."""
    L, P, M, D, B = len(shapes_l), 4, 8, 32, 2
    S = sum(h * w for h, w in shapes_l)
    gen = torch.Generator().manual_seed(5)
    ref1 = torch.cat([torch.stack(torch.meshgrid((torch.arange(h) + 0.5) / h, (torch.arange(w) + 0.5) / w, indexing="ij"), -1).reshape(-1, 2).flip(-1)
                      for h, w in shapes_l])                                                    # (S, 2) as (x, y)
    ref = ref1[None, :, None, :].expand(B if ref_batch else 1, S, L, 2).contiguous()
    if ref_batch:
        ref = ref + 0.01 * torch.randn(ref.shape, generator=gen)
    ncol = 3 * M * L * P
    ld = ncol + 8                                                                               # a padded projection row
    offaw_full = torch.zeros(B * S, ld)
    offaw_full[:, : 2 * M * L * P] = (torch.rand(B * S, 2 * M * L * P, generator=gen) * 2 - 1) * spread
    offaw_full[:, 2 * M * L * P: ncol] = torch.randn(B * S, M * L * P, generator=gen) * 2
    offaw = offaw_full.cuda()[:, :ncol]                                                         # (B * S, ncol) view with stride ld
    value = torch.randn(B, S, M, D, generator=gen).to(torch.bfloat16).cuda()
    go = torch.randn(B, S, M * D, generator=gen).to(torch.bfloat16).cuda()
    shapes = torch.tensor(shapes_l, dtype=torch.int64).cuda()
    start = torch.cat([shapes.new_zeros(1), (shapes[:, 0] * shapes[:, 1]).cumsum(0)[:-1]])
    refd = ref.cuda()
    assert K.msdeform_fused_available(shapes_l, B, M, D, L, S, P)
    # the pair
    loc, aw = K.msda_prep_fwd(offaw, refd, shapes, B, S, M, L, P)
    out0 = K.msdeform_attn_fwd(value, shapes, start, loc, aw, out_dtype=torch.bfloat16)
    gv0, gl0, ga0 = K.msdeform_attn_bwd(value, shapes, start, loc, aw, go, shapes_host=shapes_l)
    d0 = K.msda_prep_bwd(gl0, ga0, aw, shapes, ncol)
    # fused
    out1 = K.msdeform_attn_fused_fwd(value, shapes, start, offaw, refd, L, P, out_dtype=torch.bfloat16)
    gv1, d1 = K.msdeform_attn_fused_bwd(value, shapes, start, offaw, refd, L, P, go, shapes_l)
    assert torch.equal(out0, out1)                                                              # the same arithmetic, term for term
    _close(gv1, gv0, 2e-5 * float(gv0.abs().max()) + 1e-6, 1e-4)
    assert tuple(d1.shape) == (B * S, ncol) and d1.dtype == torch.bfloat16
    den = float(d0.float().abs().max())
    assert float((d1.float() - d0.float()).abs().max()) <= 2 ** -7 * den                       # one bf16 ulp of the largest entries
    assert float((d1.float() - d0.float()).norm() / d0.float().norm()) < 2e-3


@pytest.mark.parametrize("shapes_l,spread,tile", [([(16, 32), (8, 16), (4, 8)], 4.0, None), ([(27, 43), (13, 22), (7, 11)], 6.0, "8,32,78"),
                                                  ([(27, 43), (13, 22), (7, 11)], 40.0, "8,16,8"), ([(9, 5)], 3.0, "4,4,1"),
                                                  ([(33, 70), (17, 35), (9, 18), (5, 9)], 5.0, "16,16,40"),
                                                  ([(4, 8), (8, 16), (16, 32)], 3.0, None), ([(7, 11), (13, 22), (27, 43)], 6.0, "8,16,30")])
def test_msdeform_fwd_tiled_equals_gather(K, shapes_l, spread, tile, monkeypatch):
    """uenc_msdeform_attn_fwd_tiled (encoder geometry: queries = the maps' pixels, value boxes staged in LDS) against the general gather
    kernel on the same inputs, fp32 and bf16 results: offsets of a few pixels (everything staged), far offsets with a tiny LDS budget (levels
    fall back to the global gather inside the kernel), samples outside the maps, odd map sizes whose levels do not nest, 1 and 4 levels,
    levels listed fine-to-coarse and coarse-to-fine (the pixel decoder's order: the regions are cut from the finest map wherever it is).
    The reference's own op is pinned by test_msdeform_golden; this pins the second kernel to the first."""
    if tile is not None:
        monkeypatch.setenv("UENC_MSDA_TILE", tile)            # region size on level 0 and LDS budget in KB (read per launch)
    L, P, M, D, B = len(shapes_l), 4, 8, 32, 2
    S = sum(h * w for h, w in shapes_l)
    gen = torch.Generator().manual_seed(11)
    ref = torch.cat([torch.stack(torch.meshgrid((torch.arange(h) + 0.5) / h, (torch.arange(w) + 0.5) / w, indexing="ij"), -1).reshape(-1, 2).flip(-1)
                     for h, w in shapes_l])                                                     # (S, 2) as (x, y): pixel centres
    norm = torch.tensor([[w, h] for h, w in shapes_l], dtype=torch.float32)                     # (L, 2)
    off = (torch.rand(B, S, M, L, P, 2, generator=gen) * 2 - 1) * spread
    loc = (ref[None, :, None, None, None, :] + off / norm[None, None, None, :, None, :]).contiguous().cuda()
    aw = torch.softmax(torch.randn(B, S, M, L * P, generator=gen), -1).view(B, S, M, L, P).contiguous().cuda()
    value = torch.randn(B, S, M, D, generator=gen).to(torch.bfloat16).cuda()
    shapes = torch.tensor(shapes_l, dtype=torch.int64).cuda()
    start = torch.cat([shapes.new_zeros(1), (shapes[:, 0] * shapes[:, 1]).cumsum(0)[:-1]])
    assert K.msdeform_tiled_eligible(value, shapes_l, S, L, P)
    for odt, tol in ((torch.float32, 2e-6), (torch.bfloat16, 2 ** -8)):
        want = K.msdeform_attn_fwd(value, shapes, start, loc, aw, out_dtype=odt)                # no shapes_host: the gather kernel
        got = K.msdeform_attn_fwd(value, shapes, start, loc, aw, out_dtype=odt, shapes_host=shapes_l)
        den = float(want.float().abs().max())
        assert float((got.float() - want.float()).abs().max()) <= tol * den, (odt, float((got.float() - want.float()).abs().max()), den)
    # not the encoder's geometry: the wrapper must keep the gather kernel (fewer queries than pixels)
    assert not K.msdeform_tiled_eligible(value, shapes_l, S - 1, L, P)
    # the fused tiled forward (locations / weights derived in the kernel from the projection row) against glue kernel + gather kernel
    Mh = M
    ncol = 3 * Mh * L * P
    offaw = torch.zeros(B * S, ncol + 4)
    offaw[:, : 2 * Mh * L * P] = (torch.rand(B * S, 2 * Mh * L * P, generator=gen) * 2 - 1) * spread
    offaw[:, 2 * Mh * L * P: ncol] = torch.randn(B * S, Mh * L * P, generator=gen) * 2
    offaw = offaw.cuda()[:, :ncol]                                                              # padded rows (ld = ncol + 4)
    refd = ref[None, :, None, :].expand(1, S, L, 2).contiguous().cuda()
    loc2, aw2 = K.msda_prep_fwd(offaw, refd, shapes, B, S, Mh, L, P)
    want = K.msdeform_attn_fwd(value, shapes, start, loc2, aw2, out_dtype=torch.bfloat16)
    got = K.msdeform_attn_fused_fwd(value, shapes, start, offaw, refd, L, P, out_dtype=torch.bfloat16, shapes_host=shapes_l)
    den = float(want.float().abs().max())
    assert float((got.float() - want.float()).abs().max()) <= 2 ** -8 * den


@pytest.mark.parametrize("shapes_l,spread,tile", [([(4, 8), (8, 16), (16, 32)], 3.0, None), ([(7, 11), (13, 22), (27, 43)], 6.0, "8,16,30"),
                                                  ([(27, 43), (13, 22), (7, 11)], 40.0, "8,24,6"), ([(9, 5)], 3.0, "4,4,1"),
                                                  ([(33, 70), (17, 35), (9, 18), (5, 9)], 5.0, None)])
def test_msdeform_fused_bwd_tiled_equals_one_kernel(K, shapes_l, spread, tile, monkeypatch):
    """The fused backward with d(offaw) from the LDS-tiled kernel + append-only binned kernel (encoder geometry) against the one-kernel form
    (UENC_MSDA_TILED_BWD=0) on the same inputs: d(offaw) within one bf16 ulp of the largest entries, grad_value within the float-atomics
    order -- staged, partially staged and unstaged levels, samples outside the maps, 1 / 3 / 4 levels in both orders, fp32 and bf16 grad_out."""
    L, P, M, D, B = len(shapes_l), 4, 8, 32, 2
    S = sum(h * w for h, w in shapes_l)
    gen = torch.Generator().manual_seed(21)
    ref1 = torch.cat([torch.stack(torch.meshgrid((torch.arange(h) + 0.5) / h, (torch.arange(w) + 0.5) / w, indexing="ij"), -1).reshape(-1, 2).flip(-1)
                      for h, w in shapes_l])
    ref = ref1[None, :, None, :].expand(1, S, L, 2).contiguous().cuda()
    ncol = 3 * M * L * P
    offaw = torch.zeros(B * S, ncol)
    offaw[:, : 2 * M * L * P] = (torch.rand(B * S, 2 * M * L * P, generator=gen) * 2 - 1) * spread
    offaw[:, 2 * M * L * P:] = torch.randn(B * S, M * L * P, generator=gen) * 2
    offaw = offaw.cuda()
    value = torch.randn(B, S, M, D, generator=gen).to(torch.bfloat16).cuda()
    shapes = torch.tensor(shapes_l, dtype=torch.int64).cuda()
    start = torch.cat([shapes.new_zeros(1), (shapes[:, 0] * shapes[:, 1]).cumsum(0)[:-1]])
    if tile is not None:
        monkeypatch.setenv("UENC_MSDA_TILE_BWD", tile)
    for godt in (torch.bfloat16, torch.float32):
        go = torch.randn(B, S, M * D, generator=gen).to(godt).cuda()
        monkeypatch.setenv("UENC_MSDA_TILED_BWD", "0")
        gv0, d0 = K.msdeform_attn_fused_bwd(value, shapes, start, offaw, ref, L, P, go, shapes_l)
        monkeypatch.setenv("UENC_MSDA_TILED_BWD", "1")
        gv1, d1 = K.msdeform_attn_fused_bwd(value, shapes, start, offaw, ref, L, P, go, shapes_l)
        _close(gv1, gv0, 2e-5 * float(gv0.abs().max()) + 1e-6, 1e-4)
        den = float(d0.float().abs().max())
        assert float((d1.float() - d0.float()).abs().max()) <= 2 ** -7 * den
        assert float((d1.float() - d0.float()).norm() / d0.float().norm()) < 2e-3


@pytest.mark.parametrize("M,N,K_", [(20000, 256, 256), (86016, 256, 1024), (50000, 192, 192), (70001, 256, 2048), (33000, 200, 320)])
def test_gemm_nt_ln_fused_epilogue(K, M, N, K_):
    """Linear -> residual add -> LayerNorm with the LayerNorm inside the GEMM epilogue (uenc_gemm_nt_ln) against uenc_gemm_nt +
    uenc_layernorm_fwd on the same operands: the half-height kernel (N = 256, K <= 1024), the 256 x 192 and 256 x 256 tiles of the persistent
    kernel, ragged M and an N that is not a multiple of 16; pre-norm sum, fp32 / bf16 normalised rows and (mean, rstd)."""
    a = _r(M, K_, seed=1, dtype=torch.bfloat16)
    w = _r(N, K_, seed=2, scale=K_ ** -0.5, dtype=torch.bfloat16)
    bias, res = _r(N, seed=3), _r(M, N, seed=4)
    gamma, beta = 1.0 + 0.3 * _r(N, seed=5), 0.2 * _r(N, seed=6)
    fz = K.gemm_nt_ln(a, w, bias, res, gamma, beta, eps=1e-5)
    assert fz is not None, "shape should take a fused kernel"
    h, y32, y16, st = fz
    h0 = K.gemm_nt(a, w, bias=bias, epilogue=K.EPI_RESIDUAL, aux=res, out_dtype=torch.float32)
    tw = []
    y0, _, st0 = K.layernorm_fwd(h0, gamma, beta, out_dtype=torch.float32, twin=tw, eps=1e-5)
    assert torch.equal(h, h0)                                   # the same GEMM arithmetic
    _close(y32, y0, 2e-5, 2e-5)
    _close(st, st0, 1e-5 * float(st0.abs().max()), 2e-5)
    assert float((y16.float() - tw[0].float()).abs().max()) <= 2 ** -7 * float(y0.abs().max())
    # shapes the fused kernels do not take are refused, nothing written
    assert K.gemm_nt_ln(a[:300], w, bias, res[:300], gamma, beta) is None
