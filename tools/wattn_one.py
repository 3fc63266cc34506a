"""One window-attention backward launch shape repeated (for rocprofv3 --pmc): stage-3 geometry of Swin-L (64 x 128 tokens, C 768, ws 12, shift 6)."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "uni-encoder-code_amd"))
from uenc import kernels as K
H, W, C = 64, 128, 768
B, ws, nH = 2, 12, C // 32
qkv = torch.randn(B, H, W, 3 * C, device="cuda").to(torch.bfloat16)
qb = torch.randn(3 * C, device="cuda").to(torch.bfloat16)
table = torch.randn((2 * ws - 1) ** 2, nH, device="cuda") * 0.5
bq, bk = K.relpos_expand(table, ws)
do = torch.randn(B, H, W, C, device="cuda").to(torch.bfloat16)
out, lse = K.window_attn_fwd(qkv, qb, bq, ws, 6, 32 ** -0.5, want_lse=True)
for _ in range(6):
    K.window_attn_bwd(qkv, qb, bq, bk, out, do, ws, 6, 32 ** -0.5, lse=lse)
torch.cuda.synchronize()
