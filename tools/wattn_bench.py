"""Window-attention kernel timing on the Swin-L stage shapes (GPU box)."""
import os, sys, torch
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

for (H, W, C, tag) in [(256, 512, 192, "s1"), (128, 256, 384, "s2"), (64, 128, 768, "s3"), (32, 64, 1536, "s4")]:
    B, ws, nH = 2, 12, C // 32
    qkv = torch.randn(B, H, W, 3 * C, device="cuda").to(torch.bfloat16)
    qb = torch.randn(3 * C, device="cuda").to(torch.bfloat16)
    table = torch.randn((2 * ws - 1) ** 2, nH, device="cuda") * 0.5
    bq, bk = K.relpos_expand(table, ws)
    do = torch.randn(B, H, W, C, device="cuda").to(torch.bfloat16)
    for shift in (0, 6):
        out, lse = K.window_attn_fwd(qkv, qb, bq, ws, shift, 32 ** -0.5, want_lse=True)
        tf = timeit(lambda: K.window_attn_fwd(qkv, qb, bq, ws, shift, 32 ** -0.5))
        tb = timeit(lambda: K.window_attn_bwd(qkv, qb, bq, bk, out, do, ws, shift, 32 ** -0.5, lse=lse))
        Hp, Wp = -(-H // ws) * ws, -(-W // ws) * ws
        nwin = B * (Hp // ws) * (Wp // ws) * nH
        gf_f = nwin * 2 * 2 * 144 * 144 * 32 / 1e9
        print(f"{tag} shift {shift}: fwd {tf*1e3:8.1f} us ({gf_f/tf:6.0f} TF/s)   bwd {tb*1e3:8.1f} us ({gf_f*3.5/tb:6.0f} TF/s)  WGs {nwin}")
