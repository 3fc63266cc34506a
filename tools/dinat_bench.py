"""DiNAT-L backbone alone, forward + backward at 1024 x 2048, batch 2 (GPU box): ms per step and the kernel split, for the
neighbourhood-attention kernels' share (not the headline benchmark: bench.py measures the Swin-L configuration)."""
import sys, time, torch
sys.path.insert(0, '/root/repo/uni-encoder-code_amd')
import model  # noqa: F401
from uenc import ops
from uenc.modeling.backbone.dinat import DiNAT

torch.manual_seed(0)
dil = [[1, 16, 1], [1, 8, 1, 8], [1, 4] * 9, [1, 2, 1, 2, 1]]          # 1 / maximum dilation that fits kernel 7 at 1/4 .. 1/32 of 1024 x 2048
m = DiNAT(embed_dim=192, mlp_ratio=2.0, depths=[3, 4, 18, 5], num_heads=[6, 12, 24, 48], kernel_size=7, dilations=dil).cuda()
m.eval()
img = torch.randn(2, 3, 1024, 2048, device="cuda")

def step():
    for p in m.parameters():
        p.grad = None
    ops.CACHE.refresh()
    outs = m(img)
    sum(o.float().square().mean() for o in outs.values()).backward()
    ops.flush_wgrads()

for _ in range(3):
    step()
torch.cuda.synchronize()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 5
t0 = time.perf_counter()
for _ in range(n):
    step()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / n
print(f"DiNAT-L backbone fwd+bwd bs 2 1024x2048: {dt*1e3:.1f} ms/step, {2/dt:.2f} img/s, params {sum(p.numel() for p in m.parameters())/1e6:.1f} M")
