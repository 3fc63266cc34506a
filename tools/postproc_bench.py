"""Inference post-processing at 1024 x 2048, Q = 150, C = 19 (GPU box): fused kernels vs the separate passes
(upsample kernel + sigmoid + einsum / argmax in ATen)."""
import sys, torch
sys.path.insert(0, '/root/repo/uni-encoder-code_amd'); sys.path.insert(0, '/root/repo')
from uenc import kernels as K
from oracle import postproc_ref as P

def timeit(fn, n=5):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n

cls, masks = P.synthetic_predictions(150, 19, 256, 512, seed=3)
cls, masks = cls.cuda(), masks.cuda()
p = torch.softmax(cls, -1)[:, :-1].contiguous()
scores, labels = torch.softmax(cls, -1).max(-1)
sc = torch.where(scores > 0.3, scores, torch.zeros_like(scores))
size = (1024, 2048)
t_sem = timeit(lambda: K.postproc_semantic(masks, p, size, size))
t_stats = timeit(lambda: K.postproc_panoptic_stats(masks, sc, size, size))
ids, counts = K.postproc_panoptic_stats(masks, sc, size, size)
segid = torch.arange(150, dtype=torch.int32, device="cuda")
t_label = timeit(lambda: K.postproc_panoptic_label(masks, ids, segid, size))
def separate_sem():
    up = K.upsample_bilinear(masks[None], size)[0]
    return torch.einsum("qc,qhw->chw", p, up.sigmoid())
def separate_pan():
    up = K.upsample_bilinear(masks[None], size)[0]
    return (sc.view(-1, 1, 1) * up.sigmoid()).argmax(0)
t_sep_sem, t_sep_pan = timeit(separate_sem), timeit(separate_pan)
print(f"semantic: fused {t_sem*1e3:.0f} us vs upsample + sigmoid + einsum {t_sep_sem*1e3:.0f} us")
print(f"panoptic: fused stats {t_stats*1e3:.0f} us + label {t_label*1e3:.0f} us vs upsample + sigmoid + argmax {t_sep_pan*1e3:.0f} us (+ 3 .item() per query in the reference)")
