// Shared between na2d.hip (VALU kernels, launch entry points) and na2d_mfma.hip (MFMA kernels for window sizes <= 7).
#pragma once
#include "common.h"

struct Na2d {
    const bf16* qkv; const float* rpb; bf16* out; float* lse;
    const bf16* dout; bf16* dqkv; float* drpb; float* delta;
    int B, H, W, nH, d;
    float scale;
    int hh_max, hw_max;      // bwd_kv tiled: LDS halo extents (class positions) for this launch
    int tiles_x, tiles_y, nt; float inv_tiles_x;     // MFMA kernels: tiles per residue class, tiles per workgroup
};

struct AxisWin { int start, r, pb0; };       // first class position of the window, residue, bias index of slot 0

__device__ __forceinline__ AxisWin axis_win(int t, int len, int d, int K) {
    AxisWin a;
    a.r = t % d;
    const int p = t / d, L = (len - a.r + d - 1) / d;
    a.start = min(max(p - K / 2, 0), L - K);
    a.pb0 = a.start - p + K - 1;
    return a;
}

// na2d_mfma.hip
int na2d_mfma_fwd(const Na2d& p, int K, hipStream_t stream);
int na2d_mfma_bwd(Na2d& p, int K, hipStream_t stream);
bool na2d_mfma_supported(int H, int W, int nH, int K, int dilation);
