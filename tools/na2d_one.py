"""One forward + backward of the neighbourhood-attention kernels at the DiNAT-L stage-1 size (for rocprofv3 counter passes)."""
import sys, torch
sys.path.insert(0, '/root/repo/uni-encoder-code_amd')
from uenc import kernels as K
B, ks, H, W, nH = 2, 7, 256, 512, 6
C = nH * 32
qkv = torch.randn(B, H, W, 3 * C, device="cuda").to(torch.bfloat16)
rpb = torch.randn(nH, 13, 13, device="cuda") * 0.5
dout = torch.randn(B, H, W, C, device="cuda").to(torch.bfloat16)
drpb = torch.zeros_like(rpb)
for _ in range(3):
    out, lse = K.na2d_fwd(qkv, rpb, nH, ks, 1, 32 ** -0.5)
    K.na2d_bwd(qkv, rpb, out, dout, lse, nH, ks, 1, 32 ** -0.5, drpb)
torch.cuda.synchronize()
