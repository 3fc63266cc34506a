"""Neighbourhood-attention kernels at the DiNAT-L stage sizes of a 1024 x 2048 image, batch 2 (GPU box): time and effective
bandwidth over the algorithmic bytes (q, k, v read + out written forward; backward: qkv, out, dout read + dqkv written)."""
import sys, torch
sys.path.insert(0, '/root/repo/uni-encoder-code_amd')
from uenc import kernels as K

def timeit(fn, n=10):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n

B, ks = 2, 7
for tag, H, W, nH, dils in [("s1", 256, 512, 6, (1, 16)), ("s2", 128, 256, 12, (1, 8)), ("s3", 64, 128, 24, (1, 4)), ("s4", 32, 64, 48, (1, 2))]:
    C = nH * 32
    qkv = torch.randn(B, H, W, 3 * C, device="cuda").to(torch.bfloat16)
    rpb = torch.randn(nH, 2 * ks - 1, 2 * ks - 1, device="cuda") * 0.5
    dout = torch.randn(B, H, W, C, device="cuda").to(torch.bfloat16)
    drpb = torch.zeros_like(rpb)
    for d in dils:
        out, lse = K.na2d_fwd(qkv, rpb, nH, ks, d, 32 ** -0.5)
        tf = timeit(lambda: K.na2d_fwd(qkv, rpb, nH, ks, d, 32 ** -0.5))
        tb = timeit(lambda: K.na2d_bwd(qkv, rpb, out, dout, lse, nH, ks, d, 32 ** -0.5, drpb))
        px = B * H * W * C
        bf, bb = px * 2 * 4, px * 2 * (3 + 1 + 1 + 3)
        fl = 2.0 * B * H * W * nH * ks * ks * 32 * 2
        print(f"{tag} {H}x{W} nH {nH:2d} d {d:2d}: fwd {tf*1e3:7.1f} us {bf/tf/1e9:5.2f} TB/s {fl/tf/1e9:5.1f} TF/s | bwd {tb*1e3:7.1f} us {bb/tb/1e9:5.2f} TB/s", flush=True)
