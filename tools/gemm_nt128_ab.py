"""A/B of the two large-tile NT kernels per workload shape: gemm_nt256 (one workgroup per CU, UENC_GEMM_VARIANT bit 262144) vs gemm_nt128 (two per CU,
bit 131072): results must be bit-identical (same k order), times per epilogue.  GPU box."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "uni-encoder-code_amd"))
from uenc import kernels as K

def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3

# (M, N, K, tag, [epilogues that occur in the workload])
shapes = [(16384, 3072, 768, "s3 fc1", ("gelu", "dgelu", "none")), (16384, 2304, 768, "s3 qkv", ("none",)), (16384, 768, 768, "s3 proj", ("res", "none")),
          (16384, 768, 3072, "s3 fc2", ("res", "none")), (16384, 768, 2304, "s3 dqkv", ("none",)),
          (4096, 1536, 6144, "s4 fc2", ("res", "none")), (4096, 6144, 1536, "s4 fc1", ("gelu", "dgelu")), (4096, 4608, 1536, "s4 qkv", ("none",)),
          (262144, 768, 192, "s1 fc1", ("gelu", "dgelu")), (262144, 192, 768, "s1 fc2", ("res", "none")), (262144, 576, 192, "s1 qkv", ("none",)),
          (262144, 192, 192, "s1 proj", ("res", "none")), (262144, 192, 576, "s1 dqkv", ("none",)),
          (65536, 1536, 384, "s2 fc1", ("gelu", "dgelu")), (65536, 384, 1536, "s2 fc2", ("res", "none")), (65536, 1152, 384, "s2 qkv", ("none",)),
          (65536, 384, 384, "s2 proj", ("res", "none")),
          (86016, 256, 1024, "enc ffn2", ("res",)), (86016, 1024, 256, "enc ffn1", ("relu", "drelu")), (86016, 256, 256, "enc proj", ("res", "none", "nonef32")),
          (86016, 288, 256, "enc offaw", ("nonef32",)), (86016, 256, 288, "enc doffaw", ("res",)),
          (262144, 256, 256, "kv proj", ("none", "nonef32")), (262144, 2304, 256, "fpn conv", ("none",)), (262144, 256, 2304, "fpn dconv", ("nonef32",)),
          (1000, 520, 256, "ragged", ("res", "gelu", "dgelu", "none", "relu", "drelu", "nonef32"))]
ok = True
print(f"{'shape':36s} {'epilogue':>9s} {'nt256 us':>9s} {'nt128 us':>9s} {'ratio':>6s}  identical")
for M, N, Kd, tag, epis in shapes:
    g = torch.Generator(device="cuda").manual_seed(M + N + Kd)
    a = torch.randn(M, Kd, device="cuda", generator=g).to(torch.bfloat16)
    w = (torch.randn(N, Kd, device="cuda", generator=g) * Kd ** -0.5).to(torch.bfloat16)
    bias = torch.randn(N, device="cuda", generator=g)
    res = torch.randn(M, N, device="cuda", generator=g)
    pre = torch.randn(M, N, device="cuda", generator=g).to(torch.bfloat16)
    for e in epis:
        def run():
            if e == "none": return (K.gemm_nt(a, w, bias=bias, out_dtype=torch.bfloat16),)
            if e == "nonef32": return (K.gemm_nt(a, w, bias=bias, out_dtype=torch.float32),)
            if e == "res": return (K.gemm_nt(a, w, bias=bias, epilogue=K.EPI_RESIDUAL, aux=res, out_dtype=torch.float32),)
            if e == "relu": return (K.gemm_nt(a, w, bias=bias, epilogue=K.EPI_RELU),)
            if e == "gelu":
                po = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
                return (K.gemm_nt(a, w, bias=bias, epilogue=K.EPI_GELU, aux_out=po), po)
            if e == "dgelu": return (K.gemm_nt(a, w, epilogue=K.EPI_MUL_DGELU, aux=pre),)
            if e == "drelu": return (K.gemm_nt(a, w, epilogue=K.EPI_MUL_DRELU, aux=pre),)
        outs, times = {}, {}
        for name, v in (("nt256", 262144), ("nt128", 131072)):
            os.environ["UENC_GEMM_VARIANT"] = str(v)
            outs[name] = [o.clone() for o in run()]
            times[name] = timeit(run)
        same = all(torch.equal(x, y) for x, y in zip(outs["nt256"], outs["nt128"]))
        ok &= same
        print(f"{tag:10s} {M:7d}x{N:5d}x{Kd:5d}   {e:>9s} {times['nt256']:9.1f} {times['nt128']:9.1f} {times['nt128'] / times['nt256']:6.2f}  {same}", flush=True)
    os.environ["UENC_GEMM_VARIANT"] = "0"
    del a, w, res, pre
print("ALL IDENTICAL" if ok else "MISMATCH")
