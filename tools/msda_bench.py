"""MSDeformAttn backward timing on the pixel-decoder encoder shape of the 1024x2048 bench (GPU box).

Offsets follow the reference's grid initialisation (direction per head, 1..P pixels); UENC_MSDA_VARIANT selects
timing-only experiments compiled into the tiled kernel."""
import math, os, sys, torch
sys.path.insert(0, '/root/repo/uni-encoder-code_amd')
from uenc import kernels as K

def timeit(fn, n=5):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n

shapes_l = [(32, 64), (64, 128), (128, 256)]
B, M, D, P = 2, 8, 32, 4
L = len(shapes_l)
S = sum(h * w for h, w in shapes_l)
ref = torch.cat([torch.stack(torch.meshgrid((torch.arange(h) + 0.5) / h, (torch.arange(w) + 0.5) / w, indexing="ij"), -1)
                 .reshape(-1, 2).flip(-1) for h, w in shapes_l])
th = torch.arange(M, dtype=torch.float32) * (2.0 * math.pi / M)
g = torch.stack([th.cos(), th.sin()], -1)
g = (g / g.abs().max(-1, keepdim=True)[0]).view(M, 1, 1, 2).repeat(1, L, P, 1)
for i in range(P):
    g[:, :, i, :] *= i + 1
jitter = float(os.environ.get("JITTER", "0.3"))
off = g[None, None] + jitter * torch.randn(B, S, M, L, P, 2)
norm = torch.tensor([[w, h] for h, w in shapes_l], dtype=torch.float32)
loc = (ref[None, :, None, None, None, :] + off / norm[None, None, None, :, None, :]).contiguous().cuda()
value = torch.randn(B, S, M, D).to(torch.bfloat16).cuda()
w = torch.rand(B, S, M, L * P).softmax(-1).view(B, S, M, L, P).contiguous().cuda()
go = torch.randn(B, S, M * D).to(torch.bfloat16).cuda()
shapes = torch.tensor(shapes_l, dtype=torch.int64).cuda()
start = torch.cat([shapes.new_zeros(1), (shapes[:, 0] * shapes[:, 1]).cumsum(0)[:-1]])
args = (value, shapes, start, loc, w, go)
t_fwd = timeit(lambda: K.msdeform_attn_fwd(value, shapes, start, loc, w, out_dtype=torch.bfloat16))
t_dir = timeit(lambda: K.msdeform_attn_bwd(*args))
t_til = timeit(lambda: K.msdeform_attn_bwd(*args, shapes_host=shapes_l))
a = K.msdeform_attn_bwd(*args)[0]; b = K.msdeform_attn_bwd(*args, shapes_host=shapes_l)[0]
print(f"jitter {jitter}: fwd {t_fwd*1e3:.0f} us  bwd direct {t_dir*1e3:.0f} us  "
      f"bwd binned {t_til*1e3:.0f} us   max|diff| {float((a - b).abs().max()):.3e} of {float(a.abs().max()):.3e}")
