"""Is the host keeping the GPU fed?  After enqueueing each phase of a step, measure how long the host then has to wait for
the GPU to drain: ~0 means the GPU had already caught up (the phase is launch-bound)."""
import sys, time, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/uni-encoder-code_amd')
import bench
from uenc import ops
from uenc.d2 import build_model
torch.manual_seed(0)
model = build_model(bench.make_cfg("cuda:0")); model.eval()
g = torch.Generator().manual_seed(1)
batch = [{"left_image": torch.randint(0, 256, (3, bench.H_IMG, bench.W_IMG), generator=g).float().cuda(), "task": "The task is panoptic",
          "type": "segmentation", "height": bench.H_IMG, "width": bench.W_IMG} for _ in range(2)]
head = model.sem_seg_head
marks = []
def mark(name):
    t = time.perf_counter(); torch.cuda.synchronize(); marks.append((name, t, time.perf_counter()))
orig_pd = head.pixel_decoder.forward_features
def pd(*a, **k):
    mark("backbone fwd enqueued")
    r = orig_pd(*a, **k); mark("pixel decoder fwd enqueued"); return r
head.pixel_decoder.forward_features = pd
for it in range(3):
    marks.clear()
    for p in model.parameters(): p.grad = None
    ops.CACHE.refresh(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    out, images = model.forward_features(batch)
    mark("decoder fwd enqueued")
    loss = bench.synthetic_loss(out)
    loss.backward()
    mark("backward enqueued")
    if it == 2:
        prev = t0
        for name, t_enq, t_done in marks:
            print(f"{name:28s} host reached +{(t_enq - prev)*1e3:7.2f} ms, then waited {(t_done - t_enq)*1e3:7.2f} ms for the GPU")
            prev = t_done
        print(f"total {(marks[-1][2] - t0)*1e3:.2f} ms (with the 4 syncs)")
