"""A/B of the dispatch for shapes with too few 256 x 256 tiles for the persistent kernel but >= 96 tiles of 128 x 256 (UENC_GEMM_VARIANT bit 2097152 = the
128 x 128 register-staged kernel as before, 0 = the half-height LDS-DMA kernel), rotating buffers."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "uni-encoder-code_amd"))
from uenc import kernels as K
shapes = [(4096, 1536, 1536, "s4 proj", ("res", "none")), (4096, 1536, 6144, "s4 fc2", ("res",)), (4096, 1536, 4608, "s4 dqkv", ("none",)),
          (16384, 256, 256, "dec kv/16", ("none", "none32")), (16384, 256, 768, "in_proj16", ("none32",)), (16384, 384, 384, "T s3 proj", ("res",)),
          (16384, 384, 1536, "T s3 fc2", ("res",)), (12288, 512, 512, "ragged", ("none", "gelu")), (12300, 264, 320, "ragged2", ("none", "dgelu"))]
NB = 6
ok = True
for M, N, Kd, tag, epis in shapes:
    a = [torch.randn(M, Kd, device="cuda").to(torch.bfloat16) for _ in range(NB)]
    w = (torch.randn(N, Kd, device="cuda") * Kd ** -0.5).to(torch.bfloat16)
    bias = torch.randn(N, device="cuda")
    for e in epis:
        od = torch.float32 if e in ("res", "none32") else torch.bfloat16
        outs = [torch.empty(M, N, device="cuda", dtype=od) for _ in range(NB)]
        if e == "res":
            aux = [torch.randn(M, N, device="cuda") for _ in range(NB)]
            fn = lambda i: K.gemm_nt(a[i], w, bias=bias, epilogue=K.EPI_RESIDUAL, aux=aux[i], out=outs[i])
        elif e == "dgelu":
            aux = [torch.randn(M, N, device="cuda").to(torch.bfloat16) for _ in range(NB)]
            fn = lambda i: K.gemm_nt(a[i], w, epilogue=K.EPI_MUL_DGELU, aux=aux[i], out=outs[i])
        elif e == "gelu":
            aux = None
            fn = lambda i: K.gemm_nt(a[i], w, bias=bias, epilogue=K.EPI_GELU, out=outs[i])
        else:
            aux = None
            fn = lambda i: K.gemm_nt(a[i], w, bias=bias, out=outs[i])
        row, ref = [], None
        for v in (2097152, 0):
            os.environ["UENC_GEMM_VARIANT"] = str(v)
            for i in range(NB): fn(i)
            torch.cuda.synchronize()
            got = outs[0].clone()
            if ref is None: ref = got
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for r in range(4):
                for i in range(NB): fn(i)
            e1.record(); torch.cuda.synchronize()
            row.append(e0.elapsed_time(e1) / (4 * NB) * 1e3)
        same = bool(torch.equal(ref, got)); close = bool(torch.allclose(ref.float(), got.float(), atol=2e-2, rtol=2e-2)); ok &= close
        tf = 2.0 * M * N * Kd / row[1] * 1e-6
        print(f"{tag:10s} {M:6d}x{N:5d}x{Kd:5d} {e:>6s}  128x128 {row[0]:7.1f}  128x256 {row[1]:7.1f} us ({tf:5.0f} TF/s)  ratio {row[1] / row[0]:.2f}  identical {same} close {close}", flush=True)
        del aux, outs
    os.environ["UENC_GEMM_VARIANT"] = "0"
    del a
print("ALL CLOSE" if ok else "MISMATCH")
