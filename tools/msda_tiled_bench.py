"""Forward of the deformable attention at the pixel decoder's size (2 x 43008 queries = pixels, 8 heads, 3 levels x 4 points): the general
gather kernel vs the LDS-tiled one, for offsets as the reference initialises them (rays of 1..4 pixels per head) and for wider random offsets."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "uni-encoder-code_amd"))
from uenc import kernels as K
shapes_l = [(32, 64), (64, 128), (128, 256)]            # coarse to fine: the pixel decoder's level order (msdeformattn.py:283-296)
B, M, D, L, P = 2, 8, 32, 3, 4
S = sum(h * w for h, w in shapes_l)
gen = torch.Generator().manual_seed(3)
ref = torch.cat([torch.stack(torch.meshgrid((torch.arange(h) + 0.5) / h, (torch.arange(w) + 0.5) / w, indexing="ij"), -1).reshape(-1, 2).flip(-1) for h, w in shapes_l])
norm = torch.tensor([[w, h] for h, w in shapes_l], dtype=torch.float32)
value = torch.randn(B, S, M, D, generator=gen).to(torch.bfloat16).cuda()
shapes = torch.tensor(shapes_l, dtype=torch.int64).cuda()
start = torch.cat([shapes.new_zeros(1), (shapes[:, 0] * shapes[:, 1]).cumsum(0)[:-1]])
aw = torch.softmax(torch.randn(B, S, M, L * P, generator=gen), -1).view(B, S, M, L, P).contiguous().cuda()
th = torch.arange(M, dtype=torch.float32) * (2.0 * torch.pi / M)
ray = torch.stack([th.cos(), th.sin()], -1)
ray = ray / ray.abs().max(-1, keepdim=True)[0]                                # (M, 2): the reference's grid_init directions
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
cases = {"init rays 1..4 px": ray[None, None, :, None, None, :] * torch.arange(1, P + 1, dtype=torch.float32)[None, None, None, None, :, None] + 0 * norm[None, None, None, :, None, :]}
for sp in (2.0, 4.0, 8.0, 16.0, 64.0):
    cases[f"uniform +-{sp:g} px"] = (torch.rand(B, S, M, L, P, 2, generator=gen) * 2 - 1) * sp
print("UENC_MSDA_TILE =", os.environ.get("UENC_MSDA_TILE", "(default 8,32,78)"))
for name, off in cases.items():
    loc = (ref[None, :, None, None, None, :] + off / norm[None, None, None, :, None, :]).expand(B, S, M, L, P, 2).contiguous().cuda()
    a = K.msdeform_attn_fwd(value, shapes, start, loc, aw, out_dtype=torch.bfloat16)
    b = K.msdeform_attn_fwd(value, shapes, start, loc, aw, out_dtype=torch.bfloat16, shapes_host=shapes_l)
    err = float((a.float() - b.float()).abs().max()) / float(a.float().abs().max())
    t0 = timeit(lambda: K.msdeform_attn_fwd(value, shapes, start, loc, aw, out_dtype=torch.bfloat16))
    t1 = timeit(lambda: K.msdeform_attn_fwd(value, shapes, start, loc, aw, out_dtype=torch.bfloat16, shapes_host=shapes_l))
    print(f"{name:22s} gather {t0:7.1f} us   tiled {t1:7.1f} us   ratio {t1 / t0:.2f}   max rel diff {err:.1e}", flush=True)
