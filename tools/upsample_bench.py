"""x4 bilinear upsample of the mask logits at the bench size: time and store bandwidth."""
import sys, torch
sys.path.insert(0, '/root/repo/uni-encoder-code_amd')
from uenc import kernels as K
x = torch.randn(2, 150, 256, 512, device="cuda")
for _ in range(3): y = K.upsample_bilinear(x, (1024, 2048))
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): y = K.upsample_bilinear(x, (1024, 2048))
e1.record(); torch.cuda.synchronize()
t = e0.elapsed_time(e1) / 10
print(f"upsample {t*1e3:.1f} us  {y.numel()*4/t/1e9:.2f} TB/s written")
z = torch.empty_like(y)
e0.record()
for _ in range(10): z.fill_(1.0)
e1.record(); torch.cuda.synchronize()
t = e0.elapsed_time(e1) / 10
print(f"fill     {t*1e3:.1f} us  {y.numel()*4/t/1e9:.2f} TB/s written")
