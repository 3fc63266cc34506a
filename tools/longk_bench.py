"""Long-contraction NT GEMM (d(mask embeddings): M = heads x queries, N = 256 channels, K = HW pixels): row-stride padding and
split-K sweep (GPU box): atomic epilogue | stored partials + sum.  Measured: padding is irrelevant, atomics cost ~10 us per split."""
import sys, torch
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


M, N, Kd = 1536, 256, 131072
fl = 2.0 * M * N * Kd / 1e9
for pad in (0, 64):
    a = torch.randn(M, Kd + pad, device="cuda").to(torch.bfloat16)[:, :Kd]
    w = torch.randn(N, Kd + pad, device="cuda").to(torch.bfloat16)[:, :Kd]
    out = torch.zeros(M, N, device="cuda")
    row = []
    for split in (16, 43, 86, 128, 256):
        t = timeit(lambda: K.gemm_nt(a, w, out=out, splitk=split))
        t2 = timeit(lambda: K.gemm_nt_splitk(a, w, split))
        row.append(f"s{split}: {t*1e3:5.0f} | {t2*1e3:4.0f} us {fl/t2:4.0f} TF")
    print(f"pad {pad:5d}  " + "  ".join(row), flush=True)
