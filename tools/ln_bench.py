"""LayerNorm forward / backward effective bandwidth on the workload's shapes (GPU box)."""
import sys, torch
sys.path.insert(0, '/root/repo/uni-encoder-code_amd')
from uenc import kernels as K

def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n

for M, C, tag in [(262144, 192, "s1"), (65536, 384, "s2"), (16384, 768, "s3"), (4096, 1536, "s4"), (86016, 256, "enc"), (300, 256, "dec")]:
    x = torch.randn(M, C, device="cuda"); dy = torch.randn(M, C, device="cuda"); dy16 = dy.to(torch.bfloat16)
    g = torch.randn(C, device="cuda"); b = torch.randn(C, device="cuda"); dres = torch.randn(M, C, device="cuda")
    dg = torch.zeros(C, device="cuda"); db = torch.zeros(C, device="cuda")
    y, h, stats = K.layernorm_fwd(x, g, b, out_dtype=torch.bfloat16)
    tf = timeit(lambda: K.layernorm_fwd(x, g, b, out_dtype=torch.bfloat16))
    bf = M * C * (4 + 2)
    t1 = timeit(lambda: K.layernorm_bwd(dy16, x, stats, g, dres=dres, dgamma=dg, dbeta=db, twin=[]))       # norm1/norm2 of a block
    b1 = M * C * (2 + 4 + 4 + 4 + 2)
    t2 = timeit(lambda: K.layernorm_bwd(dy, x, stats, g, dgamma=dg, dbeta=db))
    b2 = M * C * (4 + 4 + 4)
    print(f"{tag:4s} {M:7d}x{C:5d}  fwd {tf*1e3:6.1f} us {bf/tf/1e9:6.0f} GB/s | bwd(bf16 dy,+res,+twin) {t1*1e3:6.1f} us {b1/t1/1e9:6.0f} GB/s | "
          f"bwd(f32) {t2*1e3:6.1f} us {b2/t2/1e9:6.0f} GB/s", flush=True)
