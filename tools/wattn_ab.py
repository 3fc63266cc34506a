"""A/B of UENC_WATTN_VARIANT switches of the window-attention backward on the Swin-L stage shapes; gradients must be identical up to the float-atomics
order of the padding-slot bias sums (dqkv itself is written without atomics: bit-identical)."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "uni-encoder-code_amd"))
from uenc import kernels as K

def timeit(fn, n=10):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3

variants = [int(v) for v in (sys.argv[1:] or ["1", "0"])]
for (H, W, C, tag) in [(256, 512, 192, "s1"), (128, 256, 384, "s2"), (64, 128, 768, "s3"), (32, 64, 1536, "s4")]:
    B, ws, nH = 2, 12, C // 32
    qkv = torch.randn(B, H, W, 3 * C, device="cuda").to(torch.bfloat16)
    qb = torch.randn(3 * C, device="cuda").to(torch.bfloat16)
    table = torch.randn((2 * ws - 1) ** 2, nH, device="cuda") * 0.5
    bq, bk = K.relpos_expand(table, ws)
    do = torch.randn(B, H, W, C, device="cuda").to(torch.bfloat16)
    for shift in (0, 6):
        out, lse = K.window_attn_fwd(qkv, qb, bq, ws, shift, 32 ** -0.5, want_lse=True)
        row, ref = [], None
        for v in variants:
            os.environ["UENC_WATTN_VARIANT"] = str(v)
            got = K.window_attn_bwd(qkv, qb, bq, bk, out, do, ws, shift, 32 ** -0.5, lse=lse)
            got = got[0] if isinstance(got, (tuple, list)) else got
            got = got.clone()
            same = True if ref is None else bool(torch.equal(ref, got))
            ref = got if ref is None else ref
            row.append((v, timeit(lambda: K.window_attn_bwd(qkv, qb, bq, bk, out, do, ws, shift, 32 ** -0.5, lse=lse)), same))
        print(f"{tag} shift {shift}: " + "   ".join(f"variant {v}: {t:8.1f} us (dqkv identical {s})" for v, t, s in row), flush=True)
os.environ["UENC_WATTN_VARIANT"] = "0"
