"""Which ATen ops one bench step still launches, and from where: a TorchDispatchMode tallies every aten op that touches a CUDA
tensor by (op, output shape, innermost uenc / bench frame), forward and backward.  Counts only (kernel times: tools/prof_step.py)."""
import os, sys, collections, traceback, torch
from torch.utils._python_dispatch import TorchDispatchMode
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "uni-encoder-code_amd"))
import bench
from uenc import ops
from uenc.d2 import build_model
from uenc.dp import GradBuckets

SKIP = ("aten.view", "aten._unsafe_view", "aten.detach", "aten.t.", "aten.transpose", "aten.permute", "aten.expand", "aten.slice", "aten.select",
        "aten.unsqueeze", "aten.squeeze", "aten.as_strided", "aten.alias", "aten.reshape", "aten.split", "aten.unbind", "aten._local_scalar",
        "aten.empty", "aten.is_", "aten.sym_", "aten.stride", "aten.size", "aten.lift_fresh", "aten.new_empty", "aten.unflatten", "aten.chunk")
tally = collections.Counter()

class Tally(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        out = func(*args, **(kwargs or {}))
        name = str(func)
        if any(name.startswith(s) for s in SKIP):
            return out
        o = out[0] if isinstance(out, (tuple, list)) and out else out
        if torch.is_tensor(o) and o.is_cuda:
            site = "<autograd engine>"
            for fr in reversed(traceback.extract_stack(limit=40)):
                if ("uni-encoder-code_amd" in fr.filename or fr.filename.endswith("bench.py")) and "aten_by_site" not in fr.filename:
                    site = f"{os.path.basename(fr.filename)}:{fr.lineno} {fr.name}"
                    break
            tally[(name.replace("aten.", ""), tuple(o.shape), str(o.dtype)[6:], site)] += 1
        return out

torch.manual_seed(0)
model = build_model(bench.make_cfg("cuda:0")); model.eval()
buckets = GradBuckets(model, bucket_mb=64.0)
g = torch.Generator().manual_seed(1000)
batch = [{"left_image": torch.randint(0, 256, (3, bench.H_IMG, bench.W_IMG), generator=g).float().cuda(), "task": "The task is panoptic",
          "type": "segmentation", "height": bench.H_IMG, "width": bench.W_IMG} for _ in range(bench.PER_GPU_BATCH)]
def step():
    buckets.zero_grad(); ops.CACHE.refresh()
    out, images = model.forward_features(batch)
    with torch.no_grad():
        model.upsample_masks(out["pred_masks"], images.tensor.shape[-2:])
    bench.synthetic_loss(out).backward(); buckets.finish()
for _ in range(2): step()
torch.cuda.synchronize()
with Tally():
    step()
torch.cuda.synchronize()
print(f"{sum(tally.values())} aten launches in one step")
by_site = collections.Counter()
for (op, shp, dt, site), n in tally.items():
    by_site[site] += n
print("---- by site")
for s, n in by_site.most_common(60):
    print(f"{n:5d}  {s}")
print("---- by (op, shape, site)")
for (op, shp, dt, site), n in sorted(tally.items(), key=lambda x: -x[1])[:120]:
    print(f"{n:5d}  {op:28s} {str(shp):28s} {dt:9s} {site}")
print("---- by bytes moved (numel x itemsize x count), top 50")
import math
def nbytes(shp, dt):
    return math.prod(shp) * {"float32": 4, "bfloat16": 2, "bool": 1, "int64": 8, "uint8": 1, "float16": 2, "int32": 4}.get(dt, 4)
for (op, shp, dt, site), n in sorted(tally.items(), key=lambda x: -nbytes(x[0][1], x[0][2]) * x[1])[:50]:
    print(f"{nbytes(shp, dt) * n / 1e6:9.1f} MB {n:4d}  {op:28s} {str(shp):28s} {dt:9s} {site}")
