"""Which fp32 -> bf16 casts a bench step launches (shape, bytes, caller): wraps uenc.kernels.cast_bf16 during the third step."""
import os, sys, traceback, collections, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "uni-encoder-code_amd"))
import bench as T
from uenc import kernels as K, ops
from uenc.d2 import build_model
from uenc.dp import GradBuckets
torch.manual_seed(0)
model = build_model(T.make_cfg("cuda")); model.eval()
buckets = GradBuckets(model)
g = torch.Generator().manual_seed(1000)
batch = [{"left_image": torch.randint(0, 256, (3, T.H_IMG, T.W_IMG), generator=g).float().cuda(), "task": "The task is panoptic", "type": "segmentation",
          "height": T.H_IMG, "width": T.W_IMG} for _ in range(T.PER_GPU_BATCH)]
log = collections.Counter()
orig = K.cast_bf16
def traced(x, *a, **k):
    fr = [f for f in traceback.extract_stack()[:-1] if "uenc" in f.filename][-3:]
    log[(tuple(x.shape), " <- ".join(f"{os.path.basename(f.filename)}:{f.lineno}" for f in fr))] += 1
    return orig(x, *a, **k)
def step():
    buckets.zero_grad(); ops.begin_step(fresh_grads=True)
    out, images = model.forward_features(batch)
    with torch.no_grad(): model.upsample_masks(out["pred_masks"], images.tensor.shape[-2:])
    T.synthetic_loss(out).backward(); buckets.finish()
for i in range(3):
    if i == 2: K.cast_bf16 = traced
    step()
torch.cuda.synchronize()
for (shape, who), n in sorted(log.items(), key=lambda kv: -kv[1] * int(torch.tensor(kv[0][0]).prod())):
    mb = int(torch.tensor(shape).prod()) * 4 / 1e6
    print(f"{n:3d} x {str(shape):28s} {mb:8.1f} MB  {who}")
