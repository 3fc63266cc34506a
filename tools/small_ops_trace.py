"""Python call sites of the ATen operators a bench step runs on the decoder's small tensors (TorchDispatchMode; no GPU profiler needed)."""
import os, sys, traceback, collections, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "uni-encoder-code_amd"))
import bench as T
from uenc import ops
from uenc.d2 import build_model
from uenc.dp import GradBuckets
from torch.utils._python_dispatch import TorchDispatchMode
torch.manual_seed(0)
model = build_model(T.make_cfg("cuda")); model.eval()
buckets = GradBuckets(model)
g = torch.Generator().manual_seed(1000)
batch = [{"left_image": torch.randint(0, 256, (3, T.H_IMG, T.W_IMG), generator=g).float().cuda(), "task": "The task is panoptic", "type": "segmentation",
          "height": T.H_IMG, "width": T.W_IMG} for _ in range(T.PER_GPU_BATCH)]
def step():
    buckets.zero_grad(); ops.begin_step(fresh_grads=True)
    out, images = model.forward_features(batch)
    with torch.no_grad(): model.upsample_masks(out["pred_masks"], images.tensor.shape[-2:])
    T.synthetic_loss(out).backward(); buckets.finish()
for _ in range(2): step()
log = collections.Counter()
SKIP = ("aten::view", "aten::_unsafe_view", "aten::reshape", "aten::detach", "aten::t", "aten::transpose", "aten::slice", "aten::select", "aten::expand", "aten::unsqueeze",
        "aten::squeeze", "aten::as_strided", "aten::alias", "aten::permute", "aten::unbind", "aten::split", "aten::empty", "aten::_local_scalar_dense", "aten::is_", "aten::size", "aten::stride")
class Mode(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = func._schema.name
        if not name.startswith(SKIP):
            ts = [a for a in args if isinstance(a, torch.Tensor)]
            if ts and ts[0].is_cuda and ts[0].numel() >= int(os.environ.get("MIN_NUMEL", "1000")) and ts[0].numel() <= int(os.environ.get("MAX_NUMEL", str(2 * 150 * 2048))):
                fr = [f for f in traceback.extract_stack()[:-1] if "/uenc/" in f.filename or f.filename.endswith("bench.py") or "autograd" in f.filename][-3:]
                log[(name + (f"[{str(ts[0].dtype)[6:]}->{str((kwargs or {}).get('dtype'))[6:]}]" if name == "aten::_to_copy" else ""), tuple(ts[0].shape), " <- ".join(f"{os.path.basename(f.filename)}:{f.lineno}" for f in reversed(fr)))] += 1
        return func(*args, **(kwargs or {}))
with Mode():
    step()
torch.cuda.synchronize()
for (name, shape, who), n in sorted(log.items(), key=lambda kv: -kv[1])[:60]:
    print(f"{n:4d} x {name:22s} {str(shape):18s} {who}")
