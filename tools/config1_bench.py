"""BASELINE configs[1]: Swin-T backbone, forward only, bs 4, 1024 x 2048, one MI355X.  Prints one JSON line (img/s, ms per forward);
kept as profiles/r02_bench_config1.json.  (A parity-test configuration, not the headline bench: tests/test_model_gpu.py::
test_swin_t_backbone_full_size_config1 pins it to the oracle.)"""
import json, os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "uni-encoder-code_amd"))
from oracle import fill
from uenc import ops
from uenc.modeling.backbone.swin import SwinTransformer

m = SwinTransformer(embed_dim=96, depths=[2, 2, 6, 2], num_heads=[3, 6, 12, 24], window_size=7)
fill.fill_module(m, "backbone.")
m = m.cuda(); m.eval()
x = (torch.randint(0, 256, (4, 3, 1024, 2048), generator=torch.Generator().manual_seed(0)).float().cuda() - 120.0) / 58.0
steps, warm = 20, 5
with torch.no_grad():
    for _ in range(warm):
        ops.CACHE.refresh(); m(x)
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
    t0 = time.perf_counter()
    for a, b in ev:
        a.record(); m(x); b.record()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
ms = sorted(a.elapsed_time(b) for a, b in ev)
gf = 389.3 * 4                                   # Swin-T 1024x2048 forward = 389.3 GF per image
print(json.dumps({"config": "BASELINE configs[1]: Swin-T backbone forward only, bs 4, 1024x2048, 1 MI355X", "value": round(4 / dt, 2), "unit": "img/s",
                  "ms_per_forward": round(dt * 1e3, 3), "event_ms": {"median": round(ms[len(ms) // 2], 3), "min": round(ms[0], 3)},
                  "steps": steps, "warmup": warm, "dtype": "bf16", "data": "synthetic", "model_tflops": round(gf / 1e3 / dt, 1)}))
