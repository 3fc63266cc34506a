"""A/B of GEMM variants inside one process (UENC_GEMM_VARIANT is re-read at every launch)."""
import os, sys, torch
sys.path.insert(0, '/root/repo/uni-encoder-code_amd')
from uenc import kernels as K
def timeit(fn, n=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
shapes = [(262144, 576, 192), (65536, 1536, 384), (16384, 3072, 768), (16384, 768, 3072), (16384, 768, 768), (4096, 4608, 1536), (8192, 8192, 8192)]
variants = sys.argv[1:] or ["0", "2"]
for M, N, Kd in shapes:
    a = torch.randn(M, Kd, device="cuda").to(torch.bfloat16); w = (torch.randn(N, Kd, device="cuda") * Kd ** -0.5).to(torch.bfloat16)
    out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16); fl = 2.0 * M * N * Kd / 1e9
    res = {v: [] for v in variants}
    res["torch"] = []
    wt = w.t().contiguous()
    for rnd in range(5):
        res["torch"].append(fl / timeit(lambda: torch.matmul(a, wt, out=out)))
        for v in variants:
            os.environ["UENC_GEMM_VARIANT"] = v
            res[v].append(fl / timeit(lambda: K.gemm_nt(a, w, out=out)))
    print(f"{M}x{N}x{Kd}: " + "  ".join(f"{v}: med {sorted(r)[2]:.0f} max {max(r):.0f}" for v, r in res.items()))

print("--- wgrad (TN): variant 0 = large-tile, 4 = 128x128 register-transposing kernel")
for M, N, Kd in [(16384, 3072, 768), (16384, 768, 3072), (16384, 768, 768), (16384, 2304, 768), (65536, 1536, 384), (262144, 576, 192), (4096, 6144, 1536)]:
    dy = torch.randn(M, N, device="cuda").to(torch.bfloat16); x = torch.randn(M, Kd, device="cuda").to(torch.bfloat16)
    dw = torch.zeros(N, Kd, device="cuda"); db = torch.zeros(N, device="cuda"); fl = 2.0 * M * N * Kd / 1e9
    res = {"0": [], "4": [], "torch": []}
    for rnd in range(5):
        for v in ("0", "4"):
            os.environ["UENC_GEMM_VARIANT"] = v
            res[v].append(fl / timeit(lambda: K.gemm_tn(dy, x, dw, db)))
        res["torch"].append(fl / timeit(lambda: torch.matmul(dy.t(), x)))
    print(f"{M}x{N}x{Kd}: " + "  ".join(f"{v}: med {sorted(r)[2]:.0f} max {max(r):.0f}" for v, r in res.items()))
