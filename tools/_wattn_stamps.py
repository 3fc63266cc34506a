import os, sys, ctypes, torch, numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "uni-encoder-code_amd"))
from uenc import kernels as K
from uenc.capi import lib
H, W, C = 64, 128, 768
B, ws, nH = 2, 12, C // 32
qkv = torch.randn(B, H, W, 3 * C, device="cuda").to(torch.bfloat16)
qb = torch.randn(3 * C, device="cuda").to(torch.bfloat16)
table = torch.randn((2 * ws - 1) ** 2, nH, device="cuda") * 0.5
bq, bk = K.relpos_expand(table, ws)
do = torch.randn(B, H, W, C, device="cuda").to(torch.bfloat16)
out = K.window_attn_fwd(qkv, qb, bq, ws, 6, 32 ** -0.5)
for _ in range(3):
    K.window_attn_bwd(qkv, qb, bq, bk, out, do, ws, 6, 32 ** -0.5)
torch.cuda.synchronize()
buf = np.zeros(4 * 8 * 9 * 12, dtype=np.uint64)
lib.uenc_debug_wattn_stamps.argtypes = [ctypes.c_void_p]
print("rc", lib.uenc_debug_wattn_stamps(buf.ctypes.data))
t = buf.reshape(4, 8, 9, 12).astype(np.int64)
# s_memtime ticks: 100 MHz constant clock on gfx9 => 10 ns per tick
for b in range(1):
    print(f"block {b}: per window (it 2..5), per wave: [wait top barrier | phase A: scores+softmax, dP/P, dQ+store | wait mid | phase B | wait end | top-of-next]  in ticks")
    for it in range(3, 5):
        for w in range(9):
            x = t[b, it, w]
            nxt = t[b, it + 1, w, 0]
            print(f"  it {it} wave {w}: topwait {x[1]-x[0]:5d} | issue {x[8]-x[1]:5d} S {x[9]-x[8]:5d} max {x[10]-x[9]:4d} exp {x[11]-x[10]:5d} sum {x[6]-x[11]:4d} dP {x[7]-x[6]:5d} dQ {x[2]-x[7]:5d} | midwait {x[3]-x[2]:5d} | B {x[4]-x[3]:5d} | endwait {x[5]-x[4]:5d} | next top {nxt-x[5]:5d} | window total {nxt-x[0]:5d}")
