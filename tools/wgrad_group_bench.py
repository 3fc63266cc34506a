"""Grouped wgrad timing for the stage-1 / stage-2 problem sets under both tile classes (GPU box)."""
import sys, torch
sys.path.insert(0, '/root/repo/uni-encoder-code_amd')
from uenc import ops
from uenc.ops import WgradQueue

def build(M, C, layers):
    probs = []
    for _ in range(layers):
        for (N, Kd) in [(3 * C, C), (C, C), (4 * C, C), (C, 4 * C)]:
            dy = torch.randn(M, N, device="cuda").to(torch.bfloat16)
            x = torch.randn(M, Kd, device="cuda").to(torch.bfloat16)
            gw = torch.zeros(N, Kd, device="cuda"); gb = torch.zeros(N, device="cuda")
            probs.append((dy, x, gw, gb))
    return probs

def descs(probs, tile, tokens):
    out, fl = [], 0.0
    for dy, x, gw, gb in probs:
        M, N, Kd = dy.shape[0], dy.shape[1], x.shape[1]
        nsplit = max(1, -(-M // tokens)); mlen = -(-(M // 64) // nsplit) * 64; nsplit = -(-M // mlen)
        tiles_k = -(-Kd // tile)
        items = -(-N // tile) * tiles_k * nsplit
        out.append((dy.data_ptr(), x.data_ptr(), gw.data_ptr(), gb.data_ptr(), dy.stride(0), x.stride(0), gw.stride(0), M, N, Kd, tiles_k, mlen, nsplit, items, 0))
        fl += 2.0 * M * N * Kd
    return out, fl

def timeit(fn, n=5):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n

import os
variants = [int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "0").split(",")]      # UENC_GEMM_VARIANT values to A/B (bit 256: two-stage TN loop)
for tag, M, C in [("s1", 262144, 192), ("s2", 65536, 384), ("s3x6", 16384, 768), ("s4", 4096, 1536)]:
    probs = build(M, C, 6 if tag == 's3x6' else 2)
    for tile, tokens in [(128, 4096), (256, 8192), (256, 16384)]:
        d, fl = descs(probs, tile, tokens)
        row = []
        for v in variants:
            os.environ["UENC_GEMM_VARIANT"] = str(v)
            t = timeit(lambda: WgradQueue.launch(tile, d, probs[0][0].device))
            row.append(f"v{v}: {t*1e3:8.1f} us {fl/t/1e9:6.0f} TF/s")
        print(f"{tag} tile {tile} tokens/item {tokens:6d}: {sum(x[13] for x in d):5d} items  " + "  ".join(row), flush=True)
    del probs
