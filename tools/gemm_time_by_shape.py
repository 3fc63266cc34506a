"""Time per NT-GEMM shape inside one real bench step: wraps kernels.gemm_nt with stream events around every launch and tallies
(M, N, K, A dtype, epilogue, out dtype) -> launches, total ms, TFLOP/s, GB/s of algorithmic bytes.  Run on the GPU box."""
import os, sys, collections, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "uni-encoder-code_amd"))
import bench
from uenc import kernels as K, ops
from uenc.d2 import build_model
from uenc.dp import GradBuckets

recs = []
orig = K.gemm_nt
def logged(a, w, **kw):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    out = orig(a, w, **kw)
    e1.record()
    recs.append(((a.shape[0], w.shape[0], a.shape[1], str(a.dtype)[6:], kw.get("epilogue", 0), str(out.dtype)[6:]), e0, e1))
    return out
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
step(); step(); recs.clear(); step()
torch.cuda.synchronize()
agg = collections.defaultdict(lambda: [0, 0.0])
for k, e0, e1 in recs:
    agg[k][0] += 1; agg[k][1] += e0.elapsed_time(e1)
tot = sum(v[1] for v in agg.values())
print(f"total gemm_nt time {tot:.2f} ms in {len(recs)} launches")
# floors per launch: MFMA = 2MNK / 2.5 PFLOP/s (dense bf16 peak); HBM = algorithmic bytes / 6.3 TB/s (what MI355X_MICROARCH.md calls
# achievable of the 8 TB/s); x = measured / max(floors): how far the shape is from whichever roof bounds it
print(f"{'M':>7} {'N':>5} {'K':>5} {'A':>8} epi {'out':>8} {'n':>4} {'ms':>7} {'us/launch':>9} {'TF/s':>7} {'GB/s':>7} {'mfma_us':>8} {'hbm_us':>7} {'bound':>5} {'x':>5}")
for k, (n, ms) in sorted(agg.items(), key=lambda x: -x[1][1])[:70]:
    M, N, Kd, ad, epi, od = k
    fl = 2.0 * M * N * Kd * n
    by = n * (M * Kd * (2 if ad == "bfloat16" else 4) + N * Kd * 2 + M * N * (2 if od == "bfloat16" else 4) + (M * N * 4 if epi == 3 else 0) + (M * N * 2 if epi in (1, 4, 5) else 0))
    us = ms * 1e3 / n
    f_m, f_h = fl / n / 2.5e15 * 1e6, by / n / 6.3e12 * 1e6
    print(f"{M:7d} {N:5d} {Kd:5d} {ad:>8} {epi:3d} {od:>8} {n:4d} {ms:7.3f} {us:9.1f} {fl / ms / 1e9:7.1f} {by / ms / 1e6:7.1f} {f_m:8.1f} {f_h:7.1f} {'mfma' if f_m > f_h else 'hbm':>5} {us / max(f_m, f_h):5.2f}")
