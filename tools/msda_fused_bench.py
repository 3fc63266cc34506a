"""Fused vs module-by-module deformable attention at the pixel decoder's full size (B 2, 1024 x 2048: levels 32x64, 64x128, 128x256; 8 heads, 4 points)."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "uni-encoder-code_amd"))
from uenc import kernels as K
shapes_l = [(32, 64), (64, 128), (128, 256)]
L, P, M, D, B = 3, 4, 8, 32, 2
S = sum(h * w for h, w in shapes_l)
g = torch.Generator(device="cuda").manual_seed(0)
ref1 = torch.cat([torch.stack(torch.meshgrid((torch.arange(h) + 0.5) / h, (torch.arange(w) + 0.5) / w, indexing="ij"), -1).reshape(-1, 2).flip(-1) for h, w in shapes_l])
ref = ref1[None, :, None, :].expand(1, S, L, 2).contiguous().cuda()
ncol = 3 * M * L * P
offaw = torch.empty(B * S, ncol, device="cuda")
offaw[:, : 2 * M * L * P] = (torch.rand(B * S, 2 * M * L * P, device="cuda", generator=g) * 2 - 1) * 4
offaw[:, 2 * M * L * P:] = torch.randn(B * S, M * L * P, device="cuda", generator=g)
value = torch.randn(B, S, M, D, device="cuda", generator=g).to(torch.bfloat16)
go = torch.randn(B, S, M * D, device="cuda", generator=g).to(torch.bfloat16)
shapes = torch.tensor(shapes_l, dtype=torch.int64).cuda()
start = torch.cat([shapes.new_zeros(1), (shapes[:, 0] * shapes[:, 1]).cumsum(0)[:-1]])
def timeit(fn, n=10):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
def pair_f():
    loc, aw = K.msda_prep_fwd(offaw, ref, shapes, B, S, M, L, P)
    return K.msdeform_attn_fwd(value, shapes, start, loc, aw, out_dtype=torch.bfloat16), loc, aw
_, loc, aw = pair_f()
def pair_b():
    gv, gl, ga = K.msdeform_attn_bwd(value, shapes, start, loc, aw, go, shapes_host=shapes_l)
    return K.msda_prep_bwd(gl, ga, aw, shapes, ncol)
print(f"forward : prep + core {timeit(lambda: pair_f()):8.1f} us   fused {timeit(lambda: K.msdeform_attn_fused_fwd(value, shapes, start, offaw, ref, L, P, out_dtype=torch.bfloat16)):8.1f} us")
print(f"backward: core + prep {timeit(pair_b):8.1f} us   fused {timeit(lambda: K.msdeform_attn_fused_bwd(value, shapes, start, offaw, ref, L, P, go, shapes_l)):8.1f} us")
os.environ["UENC_MSDA_TILED_BWD"] = "0"
t0 = timeit(lambda: K.msdeform_attn_fused_bwd(value, shapes, start, offaw, ref, L, P, go, shapes_l))
os.environ["UENC_MSDA_TILED_BWD"] = "1"
t1 = timeit(lambda: K.msdeform_attn_fused_bwd(value, shapes, start, offaw, ref, L, P, go, shapes_l))
print(f"fused backward: one gather kernel {t0:8.1f} us   LDS-tiled gather + append-only bins {t1:8.1f} us")
