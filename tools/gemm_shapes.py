"""Which NT GEMM shapes does one bench step launch?  Wraps kernels.gemm_nt and tallies (M, N, K, A dtype, epilogue, out dtype)."""
import os, sys, collections, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "uni-encoder-code_amd"))
import bench
from uenc import kernels as K, ops
from uenc.d2 import build_model
from uenc.dp import GradBuckets

tally = collections.Counter()
orig = K.gemm_nt
def logged(a, w, **kw):
    out = kw.get("out")
    od = (out.dtype if out is not None else kw.get("out_dtype", torch.bfloat16))
    tally[(a.shape[0], w.shape[0], a.shape[1], str(a.dtype)[6:], kw.get("epilogue", 0), str(od)[6:], kw.get("splitk", 1))] += 1
    return orig(a, w, **kw)
K.gemm_nt = logged
torch.manual_seed(0)
model = build_model(bench.make_cfg("cuda:0")); model.eval()
buckets = GradBuckets(model, bucket_mb=64.0)
g = torch.Generator().manual_seed(1000)
batch = [{"left_image": torch.randint(0, 256, (3, bench.H_IMG, bench.W_IMG), generator=g).float().cuda(), "task": "The task is panoptic",
          "type": "segmentation", "height": bench.H_IMG, "width": bench.W_IMG} for _ in range(bench.PER_GPU_BATCH)]
def step():
    buckets.zero_grad(); ops.CACHE.refresh()
    out, images = model.forward_features(batch)
    bench.synthetic_loss(out).backward(); buckets.finish()
step(); tally.clear(); step()
torch.cuda.synchronize()
print("M, N, K, A dtype, epilogue, out dtype, splitk : launches   (fp32-A or not-big shapes with M >= 4096)")
for k, n in sorted(tally.items(), key=lambda x: -x[0][0] * x[0][1] * x[0][2] * x[1]):
    M, N, Kd, ad = k[:4]
    big = ad == "bfloat16" and Kd % 64 == 0 and Kd >= 128 and N >= 192 and N % 8 == 0 and ((M + 255) // 256) * ((N + 255) // 256) >= 160
    if not big and M >= 4096:
        print(k, n)
