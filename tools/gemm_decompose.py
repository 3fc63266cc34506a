"""Main loop vs epilogue per NT-256 shape: UENC_GEMM_VARIANT 0 (normal), 16384 (no epilogue at all), 8192 (staged epilogue without global stores / aux loads;
fp32 outputs only).  Timing only -- the variants' results are wrong by construction."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "uni-encoder-code_amd"))
from uenc import kernels as K

def timeit(fn, n=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3

shapes = [(16384, 3072, 768, "s3 fc1"), (16384, 2304, 768, "s3 qkv"), (16384, 768, 768, "s3 proj"), (16384, 768, 3072, "s3 fc2"), (16384, 768, 2304, "s3 dqkv"),
          (4096, 1536, 6144, "s4 fc2"), (4096, 6144, 1536, "s4 fc1"), (4096, 4608, 1536, "s4 qkv"),
          (262144, 768, 192, "s1 fc1"), (262144, 192, 768, "s1 fc2"), (65536, 1536, 384, "s2 fc1"), (86016, 256, 1024, "enc ffn2"), (86016, 1024, 256, "enc ffn1")]
print(f"{'shape':34s} {'epilogue':>13s} {'normal':>9s} {'no-epi':>9s} {'no-store':>9s}   mfma_floor hbm_floor")
for M, N, Kd, tag in shapes:
    a = torch.randn(M, Kd, device="cuda").to(torch.bfloat16)
    w = (torch.randn(N, Kd, device="cuda") * Kd ** -0.5).to(torch.bfloat16)
    bias = torch.randn(N, device="cuda")
    out16 = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    res = torch.randn(M, N, device="cuda")
    pre = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    runs = {"none bf16": (lambda: K.gemm_nt(a, w, bias=bias, out=out16), M * Kd * 2 + N * Kd * 2 + M * N * 2),
            "residual f32": (lambda: K.gemm_nt(a, w, bias=bias, epilogue=K.EPI_RESIDUAL, aux=res, out=res), M * Kd * 2 + N * Kd * 2 + M * N * 8),
            "gelu": (lambda: K.gemm_nt(a, w, bias=bias, epilogue=K.EPI_GELU, aux_out=pre, out=out16), M * Kd * 2 + N * Kd * 2 + M * N * 4),
            "mul_dgelu": (lambda: K.gemm_nt(a, w, epilogue=K.EPI_MUL_DGELU, aux=pre, out=out16), M * Kd * 2 + N * Kd * 2 + M * N * 4)}
    for name, (fn, by) in runs.items():
        row = []
        for v in (0, 16384, 8192):
            os.environ["UENC_GEMM_VARIANT"] = str(v)
            row.append(timeit(fn))
        print(f"{tag:10s} {M:7d}x{N:5d}x{Kd:5d} {name:>13s} {row[0]:9.1f} {row[1]:9.1f} {row[2]:9.1f}   {2.0 * M * N * Kd / 2.5e15 * 1e6:8.1f} {by / 6.3e12 * 1e6:8.1f}", flush=True)
    os.environ["UENC_GEMM_VARIANT"] = "0"
    del a, w, out16, res, pre
