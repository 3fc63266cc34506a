"""A few grouped-wgrad launches of one problem set for rocprofv3 --pmc passes (FETCH_SIZE: L2 <-> fabric read traffic).
usage: wgrad_pmc.py <UENC_GEMM_VARIANT> [set]    set: s3 (default) | s1 | s4"""
import os, sys, torch
os.environ["UENC_GEMM_VARIANT"] = sys.argv[1] if len(sys.argv) > 1 else "0"
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "uni-encoder-code_amd"))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from uenc.ops import WgradQueue
import importlib.util
which = sys.argv[2] if len(sys.argv) > 2 else "s3"
M, C, layers = {"s3": (16384, 768, 6), "s1": (262144, 192, 2), "s4": (4096, 1536, 2)}[which]
probs = []
for _ in range(layers):
    for (N, Kd) in [(3 * C, C), (C, C), (4 * C, C), (C, 4 * C)]:
        probs.append((torch.randn(M, N, device="cuda").to(torch.bfloat16), torch.randn(M, Kd, device="cuda").to(torch.bfloat16),
                      torch.zeros(N, Kd, device="cuda"), torch.zeros(N, device="cuda")))
tile, tokens = 256, 8192
d, fl, by = [], 0.0, 0.0
for dy, x, gw, gb in probs:
    Mm, N, Kd = dy.shape[0], dy.shape[1], x.shape[1]
    nsplit = max(1, -(-Mm // tokens)); mlen = -(-(Mm // 64) // nsplit) * 64; nsplit = -(-Mm // mlen)
    tiles_k = -(-Kd // tile)
    d.append((dy.data_ptr(), x.data_ptr(), gw.data_ptr(), gb.data_ptr(), dy.stride(0), x.stride(0), gw.stride(0), Mm, N, Kd, tiles_k, mlen, nsplit,
              -(-N // tile) * tiles_k * nsplit, 0))
    fl += 2.0 * Mm * N * Kd; by += 2.0 * Mm * (N + Kd)
for _ in range(3):
    WgradQueue.launch(tile, d, probs[0][0].device)
torch.cuda.synchronize()
print(f"set {which}: algorithmic operand bytes per launch {by / 1e6:.1f} MB, flops {fl / 1e12:.3f} TF")
