"""A few NT GEMM launches for rocprofv3 --pmc passes (SQ busy / wait counters of the 256-tile kernel)."""
import sys, torch
sys.path.insert(0, '/root/repo/uni-encoder-code_amd')
from uenc import kernels as K
for M, N, Kd in [(8192, 8192, 8192), (16384, 2304, 768), (262144, 576, 192)]:
    a = torch.randn(M, Kd, device="cuda").to(torch.bfloat16)
    w = (torch.randn(N, Kd, device="cuda") * Kd ** -0.5).to(torch.bfloat16)
    out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    for _ in range(3):
        K.gemm_nt(a, w, out=out)
    torch.cuda.synchronize()
