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
out, lse = K.window_attn_fwd(qkv, qb, bq, ws, 6, 32 ** -0.5, want_lse=True)
for _ in range(3):
    K.window_attn_bwd(qkv, qb, bq, bk, out, do, ws, 6, 32 ** -0.5, lse=lse)
torch.cuda.synchronize()
buf = np.zeros(4 * 8 * 10 * 12, dtype=np.uint64)
lib.uenc_debug_wattn_stamps.argtypes = [ctypes.c_void_p]
print("rc", lib.uenc_debug_wattn_stamps(buf.ctypes.data))
t = buf.reshape(4, 8, 10, 12).astype(np.int64)
for it in range(3, 5):
    for w in range(10):
        x = t[0, it, w]; nxt = t[0, it + 1, w, 0]
        print(f"  it {it} wave {w}: dmawait {x[0]-x[8]:5d} topwait {x[1]-x[0]:5d} | S {x[9]-x[1]:5d} exp {x[6]-x[9]:5d} dP {x[7]-x[6]:5d} dQ {x[2]-x[7]:5d} (A total {x[2]-x[1]:5d}) | midwait {x[3]-x[2]:5d} | B {x[4]-x[3]:5d} | endwait {x[5]-x[4]:5d} | next top {nxt-x[5]:5d} | window {nxt-x[0]:5d}")
