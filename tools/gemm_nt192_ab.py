"""A/B of the tile width of gemm_nt256 per workload shape: 256 x 256 (UENC_GEMM_VARIANT bit 8388608) vs 256 x 192 (bit 4194304), both with bit 262144
(no half-height kernel): results must be bit-identical (same k order), times per epilogue, plus what the dispatch heuristic picks.  GPU box."""
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
          (40008, 520, 256, "ragged", ("res", "res16", "gelu", "dgelu", "none", "relu", "drelu", "nonef32")), (99992, 200, 128, "ragged2", ("res", "dgelu", "none", "nonef32")),
          (41000, 776, 320, "ragged3", ("res", "gelu", "dgelu", "none"))]
ok = True
print(f"{'shape':36s} {'epilogue':>9s} {'256w us':>9s} {'192w us':>9s} {'ratio':>6s}  identical")
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
            if e == "res16": return (K.gemm_nt(a, w, bias=bias, epilogue=K.EPI_RESIDUAL, aux=res, out_dtype=torch.bfloat16),)
            if e == "res": return (K.gemm_nt(a, w, bias=bias, epilogue=K.EPI_RESIDUAL, aux=res, out_dtype=torch.float32),)
            if e == "relu": return (K.gemm_nt(a, w, bias=bias, epilogue=K.EPI_RELU),)
            if e == "gelu":
                po = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
                return (K.gemm_nt(a, w, bias=bias, epilogue=K.EPI_GELU, aux_out=po), po)
            if e == "dgelu": return (K.gemm_nt(a, w, epilogue=K.EPI_MUL_DGELU, aux=pre),)
            if e == "drelu": return (K.gemm_nt(a, w, epilogue=K.EPI_MUL_DRELU, aux=pre),)
        outs, times = {}, {}
        for name, v in (("nt256", 262144 | 8388608), ("nt128", 262144 | 4194304), ("auto", 0)):
            os.environ["UENC_GEMM_VARIANT"] = str(v)
            outs[name] = [o.clone() for o in run()]
            times[name] = timeit(run)
        same = all(torch.equal(x, y) for x, y in zip(outs["nt256"], outs["nt128"])) and all(torch.equal(x, y) for x, y in zip(outs["nt256"], outs["auto"]))
        ok &= same
        print(f"{tag:10s} {M:7d}x{N:5d}x{Kd:5d}   {e:>9s} {times['nt256']:9.1f} {times['nt128']:9.1f} {times['nt128'] / times['nt256']:6.2f}  {same}  auto {times['auto']:7.1f}", flush=True)
    os.environ["UENC_GEMM_VARIANT"] = "0"
    del a, w, res, pre
print("ALL IDENTICAL" if ok else "MISMATCH")
