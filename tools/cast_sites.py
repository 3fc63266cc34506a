"""Who still launches cast_f32_bf16 / cast_transpose in one bench step: wraps kernels.cast_bf16 / cast_transpose_bf16 and tallies
(shape, innermost uenc frame).  GPU box."""
import os, sys, collections, traceback, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "uni-encoder-code_amd"))
import bench
from uenc import kernels as K, ops
from uenc.d2 import build_model
from uenc.dp import GradBuckets
tally = collections.Counter()
def wrap(name):
    orig = getattr(K, name)
    def f(src, out=None):
        site = "?"
        for fr in reversed(traceback.extract_stack(limit=12)[:-1]):
            if "uni-encoder-code_amd" in fr.filename and "kernels.py" not in fr.filename:
                site = f"{os.path.basename(fr.filename)}:{fr.lineno} {fr.name}"; break
        tally[(name, tuple(src.shape), site)] += 1
        return orig(src, out) if out is not None else orig(src)
    setattr(K, name, f)
wrap("cast_bf16"); wrap("cast_transpose_bf16")
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
step(); step(); tally.clear(); step(); torch.cuda.synchronize()
for (name, shp, site), n in sorted(tally.items(), key=lambda x: -x[1] * (x[0][1][0] if x[0][1] else 1)):
    print(f"{n:4d}  {name:22s} {str(shp):22s} {site}")
