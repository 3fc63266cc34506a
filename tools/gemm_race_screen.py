"""Race screen of the 256-tile NT kernel: every shape is run `reps` times on fresh random operands while another stream keeps the
memory system busy; each result must equal the fp32 product of the same bf16 operands (to accumulation order) and be bit-identical
between repeats of the same operands.  UENC_GEMM_VARIANT selects the main loop under test."""
import os, sys, torch
sys.path.insert(0, '/root/repo/uni-encoder-code_amd')
from uenc import kernels as K

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 6
shapes = []
for Kd in (128, 192, 256, 320, 384, 512, 768, 1536, 4096):
    for (M, N) in ((1024, 1024), (4096, 2304), (8192 + 40, 1536 + 8), (16384, 768), (65536, 256), (5000, 200)):
        shapes.append((M, N, Kd))
side = torch.cuda.Stream()
junk = torch.empty(64 << 20, device="cuda")
bad = 0
for M, N, Kd in shapes:
    for r in range(reps):
        a = torch.randn(M, Kd, device="cuda").to(torch.bfloat16)
        w = torch.randn(N, Kd, device="cuda").to(torch.bfloat16)
        with torch.cuda.stream(side):
            for _ in range(4):
                junk.add_(1.0)
        o1 = K.gemm_nt(a, w, out_dtype=torch.float32)
        o2 = K.gemm_nt(a, w, out_dtype=torch.float32)
        want = a.float() @ w.float().t()
        err = float((o1 - want).abs().max() / (want.abs().max() + 1e-9))
        same = bool(torch.equal(o1, o2))
        if err > 2e-3 or not same:
            bad += 1
            print(f"BAD {M}x{N}x{Kd} rep {r}: rel-max err {err:.3e} identical {same}", flush=True)
    print(f"{M}x{N}x{Kd} ok", flush=True)
torch.cuda.synchronize()
print("race screen:", "FAILED %d" % bad if bad else "clean", f"({len(shapes)} shapes x {reps} reps, variant {os.environ.get('UENC_GEMM_VARIANT', '0')})")
sys.exit(1 if bad else 0)
