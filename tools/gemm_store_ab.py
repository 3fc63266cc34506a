"""A/B of store policies in the register-direct epilogue (UENC_GEMM_VARIANT bit 524288 = non-temporal stores) with ROTATING output buffers (1.2 GB in
flight: the Infinity Cache cannot absorb the writes, as in the real step)."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "uni-encoder-code_amd"))
from uenc import kernels as K
shapes = [(16384, 3072, 768, "s3 fc1"), (16384, 2304, 768, "s3 qkv"), (16384, 768, 3072, "s3 fc2"), (65536, 1536, 384, "s2 fc1"), (262144, 768, 192, "s1 fc1")]
NB = 6
for M, N, Kd, tag in shapes:
    a = [torch.randn(M, Kd, device="cuda").to(torch.bfloat16) for _ in range(NB)]
    w = (torch.randn(N, Kd, device="cuda") * Kd ** -0.5).to(torch.bfloat16)
    bias = torch.randn(N, device="cuda")
    outs = [torch.empty(M, N, device="cuda", dtype=torch.bfloat16) for _ in range(NB)]
    pres = [torch.randn(M, N, device="cuda").to(torch.bfloat16) for _ in range(NB)]
    runs = {"none": lambda i: K.gemm_nt(a[i], w, bias=bias, out=outs[i]),
            "gelu": lambda i: K.gemm_nt(a[i], w, bias=bias, epilogue=K.EPI_GELU, aux_out=pres[i], out=outs[i]),
            "dgelu": lambda i: K.gemm_nt(a[i], w, epilogue=K.EPI_MUL_DGELU, aux=pres[i], out=outs[i])}
    for name, fn in runs.items():
        row = []
        for v in (0, 524288, 16384):
            os.environ["UENC_GEMM_VARIANT"] = str(v)
            for i in range(NB): fn(i)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for r in range(4):
                for i in range(NB): fn(i)
            e1.record(); torch.cuda.synchronize()
            row.append(e0.elapsed_time(e1) / (4 * NB) * 1e3)
        print(f"{tag:8s} {M:7d}x{N:5d}x{Kd:5d} {name:>6s}  default {row[0]:7.1f}  nt-stores {row[1]:7.1f}  no-epilogue {row[2]:7.1f}", flush=True)
    os.environ["UENC_GEMM_VARIANT"] = "0"
    del a, outs, pres
