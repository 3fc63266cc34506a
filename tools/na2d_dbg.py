"""Phase timing of the MFMA neighbourhood-attention forward (UENC_NA2D_DEBUG: 1 skips the staging, 2 the arithmetic)."""
import os, sys, torch
sys.path.insert(0, '/root/repo/uni-encoder-code_amd')
from uenc import kernels as K
def timeit(fn, n=10):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
B, ks, H, W, nH = 2, 7, 256, 512, 6
C = nH * 32
qkv = torch.randn(B, H, W, 3 * C, device="cuda").to(torch.bfloat16)
rpb = torch.randn(nH, 13, 13, device="cuda") * 0.5
for dbg in (0, 1, 2, 3):
    os.environ["UENC_NA2D_DEBUG"] = str(dbg)
    print("dbg", dbg, f"{timeit(lambda: K.na2d_fwd(qkv, rpb, nH, ks, 1, 32 ** -0.5)) * 1e3:.1f} us", flush=True)
