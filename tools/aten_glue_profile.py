"""Which torch (aten) kernels remain in the bench step and where they come from: torch.profiler over one step, CUDA time by
(op, input shapes) with the innermost uenc/ python frame."""
import os, sys, collections, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "uni-encoder-code_amd"))
import bench
from uenc import ops
from uenc.d2 import build_model
from uenc.dp import GradBuckets
from torch.profiler import profile, ProfilerActivity

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
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof:
    step(); torch.cuda.synchronize()
agg = collections.defaultdict(lambda: [0.0, 0])
for e in prof.events():
    if not e.name.startswith("aten::") or e.device_time <= 0 or e.cpu_children and any(c.device_time > 0 for c in e.cpu_children):
        continue
    where = "?"
    for fr in (e.stack or []):
        if "uenc/" in fr or "bench.py" in fr or "/model/" in fr:
            where = fr.split("uni-encoder-code_amd/")[-1][:70]; break
    key = (e.name, str(e.input_shapes)[:60], where)
    agg[key][0] += e.device_time; agg[key][1] += 1
tot = sum(v[0] for v in agg.values())
print(f"aten leaf ops with device time: {tot/1e3:.2f} ms")
for k, v in sorted(agg.items(), key=lambda x: -x[1][0])[:70]:
    print(f"{v[0]/1e3:7.3f} ms {v[1]:4d}x  {k[0]:22s} {k[1]:60s} {k[2]}")
