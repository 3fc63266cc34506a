"""A/B of the epilogues' aux-load placement (UENC_GEMM_VARIANT bit 1048576 = per-row loads as before, 0 = requested ahead of the stores) on the workload's
shapes with ROTATING buffers (the Infinity Cache cannot hold them, as in the real step); results must be bit-identical."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "uni-encoder-code_amd"))
from uenc import kernels as K
shapes = [(16384, 3072, 768, "s3 fc1", ("dgelu",)), (16384, 768, 3072, "s3 fc2", ("res",)), (16384, 768, 768, "s3 proj", ("res",)),
          (65536, 1536, 384, "s2 fc1", ("dgelu",)), (65536, 384, 1536, "s2 fc2", ("res",)), (262144, 768, 192, "s1 fc1", ("dgelu",)),
          (262144, 192, 768, "s1 fc2", ("res",)), (262144, 192, 192, "s1 proj", ("res",)),
          (86016, 1024, 256, "enc ffn1", ("drelu",)), (86016, 256, 1024, "enc ffn2", ("res",)), (86016, 256, 256, "enc proj", ("res",)),
          (4096, 6144, 1536, "s4 fc1", ("dgelu",)), (1000, 520, 256, "ragged", ("res", "dgelu", "drelu"))]
NB = 5
ok = True
for M, N, Kd, tag, epis in shapes:
    a = [torch.randn(M, Kd, device="cuda").to(torch.bfloat16) for _ in range(NB)]
    w = (torch.randn(N, Kd, device="cuda") * Kd ** -0.5).to(torch.bfloat16)
    bias = torch.randn(N, device="cuda")
    for e in epis:
        if e in ("res", "res16"):
            aux = [torch.randn(M, N, device="cuda") for _ in range(NB)]
            od = torch.float32 if e == "res" else torch.bfloat16
            outs = [torch.empty(M, N, device="cuda", dtype=od) for _ in range(NB)]
            fn = lambda i: K.gemm_nt(a[i], w, bias=bias, epilogue=K.EPI_RESIDUAL, aux=aux[i], out=outs[i])
        else:
            aux = [torch.randn(M, N, device="cuda").to(torch.bfloat16) for _ in range(NB)]
            outs = [torch.empty(M, N, device="cuda", dtype=torch.bfloat16) for _ in range(NB)]
            ep = K.EPI_MUL_DGELU if e == "dgelu" else K.EPI_MUL_DRELU
            fn = lambda i: K.gemm_nt(a[i], w, epilogue=ep, aux=aux[i], out=outs[i])
        row, ref = [], None
        for v in (1048576, 0):
            os.environ["UENC_GEMM_VARIANT"] = str(v)
            for i in range(NB): fn(i)
            torch.cuda.synchronize()
            got = outs[0].clone()
            if ref is None: ref = got
            else: ok &= bool(torch.equal(ref, got))
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for r in range(4):
                for i in range(NB): fn(i)
            e1.record(); torch.cuda.synchronize()
            row.append(e0.elapsed_time(e1) / (4 * NB) * 1e3)
        print(f"{tag:8s} {M:7d}x{N:5d}x{Kd:5d} {e:>6s}  per-row loads {row[0]:7.1f}  ahead {row[1]:7.1f}  ratio {row[1] / row[0]:.2f}  identical {torch.equal(ref, got)}", flush=True)
        del aux, outs
    os.environ["UENC_GEMM_VARIANT"] = "0"
    del a
print("ALL IDENTICAL" if ok else "MISMATCH")
