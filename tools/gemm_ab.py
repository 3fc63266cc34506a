"""A/B of UENC_GEMM_VARIANT bits on the workload's NT shapes inside ONE process (the library reads the variable per call)."""
import os, sys, torch
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

shapes = [
    (262144, 576, 192, "s1 qkv"), (262144, 192, 192, "s1 proj"), (262144, 768, 192, "s1 fc1"), (262144, 192, 768, "s1 fc2"),
    (65536, 1152, 384, "s2 qkv"), (65536, 1536, 384, "s2 fc1"), (65536, 384, 1536, "s2 fc2"),
    (16384, 2304, 768, "s3 qkv"), (16384, 768, 768, "s3 proj"), (16384, 3072, 768, "s3 fc1"), (16384, 768, 3072, "s3 fc2"),
    (4096, 4608, 1536, "s4 qkv"), (4096, 6144, 1536, "s4 fc1"), (4096, 1536, 6144, "s4 fc2"),
    (86016, 256, 256, "enc proj"), (86016, 1024, 256, "enc ffn1"), (86016, 256, 1024, "enc ffn2"),
    (262144, 256, 256, "kv proj"), (8192, 8192, 8192, "square 8k"),
]
variants = [int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "0,128").split(",")]      # bit 128: two-stage loop, 512: no 192-wide tiles
print(f"{'shape':28s} " + " ".join(f"{'v' + str(v) + ' bf16/res/gelu/dgelu':>30s}" for v in variants) + "   (TFLOP/s)")
for M, N, Kd, tag in shapes:
    a = torch.randn(M, Kd, device="cuda").to(torch.bfloat16)
    w = (torch.randn(N, Kd, device="cuda") * Kd ** -0.5).to(torch.bfloat16)
    bias = torch.randn(N, device="cuda")
    fl = 2.0 * M * N * Kd / 1e9
    out16 = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    res = torch.randn(M, N, device="cuda")
    pre = torch.randn(M, N, device="cuda").to(torch.bfloat16)
    cols = []
    for v in variants:
        os.environ["UENC_GEMM_VARIANT"] = str(v)
        t1 = timeit(lambda: K.gemm_nt(a, w, bias=bias, out=out16))
        t2 = timeit(lambda: K.gemm_nt(a, w, bias=bias, epilogue=K.EPI_RESIDUAL, aux=res, out=res))
        t3 = timeit(lambda: K.gemm_nt(a, w, bias=bias, epilogue=K.EPI_GELU, aux_out=pre, out=out16))
        t4 = timeit(lambda: K.gemm_nt(a, w, epilogue=K.EPI_MUL_DGELU, aux=pre, out=out16))
        cols.append(f"{fl/t1:7.0f}{fl/t2:7.0f}{fl/t3:7.0f}{fl/t4:7.0f}  ")
    print(f"{tag:10s} {M:7d}x{N:5d}x{Kd:5d} " + " ".join(cols), flush=True)
    del a, w, out16, res, pre
