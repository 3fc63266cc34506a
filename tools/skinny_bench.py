"""The transformer decoder's few-tile GEMMs (M = 300 rows): K-split skinny kernel vs the 128-tile kernel (UENC_GEMM_VARIANT bit 1024 = off)."""
import os, sys, torch
sys.path.insert(0, '/root/repo/uni-encoder-code_amd')
from uenc import kernels as K

def timeit(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3

for M, N, Kd, tag in [(300, 256, 256, "proj"), (300, 256, 2048, "ffn2"), (300, 2048, 256, "ffn1"), (300, 512, 256, "qk"), (300, 20, 256, "cls"),
                      (298, 256, 256, "cls-tr"), (300, 256, 1024, "k1024")]:
    a = torch.randn(M, Kd, device="cuda").to(torch.bfloat16); a32 = a.float()
    w = (torch.randn(N, Kd, device="cuda") * Kd ** -0.5).to(torch.bfloat16)
    bias = torch.randn(N, device="cuda"); res = torch.randn(M, N, device="cuda")
    o16 = torch.empty(M, N, device="cuda", dtype=torch.bfloat16); o32 = torch.empty(M, N, device="cuda")
    row = []
    for v in (1024, 0):
        os.environ["UENC_GEMM_VARIANT"] = str(v)
        t1 = timeit(lambda: K.gemm_nt(a, w, bias=bias, out=o16))
        t2 = timeit(lambda: K.gemm_nt(a, w, bias=bias, epilogue=K.EPI_RESIDUAL, aux=res, out=o32))
        t3 = timeit(lambda: K.gemm_nt(a32, w, out=o32))
        row.append(f"{'128-tile' if v else 'skinny  '} bf16 {t1:5.1f} res {t2:5.1f} f32A {t3:5.1f} us")
    print(f"{tag:7s} {M}x{N}x{Kd:5d}  " + " | ".join(row), flush=True)
