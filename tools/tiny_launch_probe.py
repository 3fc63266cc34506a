"""Kernel-duration floor of tiny launches (run under rocprofv3 --kernel-trace --stats): a 1 K-element cast, an in-place add, a
300-row LayerNorm and a 300 x 256 x 256 GEMM, 50 times each.  Measured: 1.2 us / 1.5 us / 3.9 us / 8.1 us."""
import sys, torch
sys.path.insert(0, '/root/repo/uni-encoder-code_amd')
from uenc import kernels as K
x = torch.randn(1024, device="cuda"); y = torch.empty(1024, device="cuda", dtype=torch.bfloat16)
a = torch.randn(300, 256, device="cuda"); g = torch.ones(256, device="cuda"); b = torch.zeros(256, device="cuda")
w = torch.randn(256, 256, device="cuda").to(torch.bfloat16); a16 = a.to(torch.bfloat16)
for _ in range(50):
    K.cast_bf16(x, out=y)
    x.add_(1.0)
    K.layernorm_fwd(a, g, b, out_dtype=torch.bfloat16)
    K.gemm_nt(a16, w)
torch.cuda.synchronize()
