// Neighbourhood attention 2-D (DiNAT): what natten.NeighborhoodAttention2D computes between its
// qkv and proj Linear layers (reference call site model/modeling/backbone/dinat.py:77-79, 94; the arithmetic is NATTEN
// 0.14.4's natten2dqkrpb + softmax + natten2dav and their backward, restated in oracle/dinat_ref.py -- parity unpinned).
//
//   out[b, y, x, h, :] = sum_{i, j < k} softmax_{ij}( scale * q[b,y,x,h] . k[b, ny(y,i), nx(x,j), h] + rpb[h, by(y)+i, bx(x)+j] )
//                                       * v[b, ny(y,i), nx(x,j), h]
//
// Neighbourhood (per axis, dilation d): index t = p * d + r sees the k members of its residue class r at class positions
// start .. start + k - 1, start = clamp(p - k/2, 0, L_r - k) (clamped inside the image), and window slot i carries the bias
// row (start + i - p) + k - 1.
//
// Layout: qkv (B, H, W, 3, nH, 32) bf16 = the output of the qkv GEMM as is; out (B, H, W, nH, 32) bf16 = the proj GEMM's
// operand; head_dim is 32 (every DiNAT variant).  Mapping: four lanes per pixel and head (8 channels = one 16-byte load
// each), 16 consecutive pixels of an image row per wave, so a wave's loads of one window slot are 16 adjacent 64-byte
// segments; the k x k scores never leave registers (online softmax, one rescale per window row).  The backward is two
// kernels of the same shape, both recomputing the probabilities from the saved log-sum-exp: per query (dq, drpb, delta) and
// per key over its inverse neighbourhood (dk, dv).  HBM traffic is q, k, v, out once each (the window re-reads hit L1 / L2):
// algorithmic bytes = 4 * B*H*W*C * 2 forward, memory-side bound.
#include "common.h"

struct Na2d {
    const bf16* qkv; const float* rpb; bf16* out; float* lse;
    const bf16* dout; bf16* dqkv; float* drpb; float* delta;
    int B, H, W, nH, d;
    float scale;
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

__device__ __forceinline__ float quad_sum(float v) {       // over the 4 lanes of a pixel
    v += __shfl_xor(v, 1);
    v += __shfl_xor(v, 2);
    return v;
}

__device__ __forceinline__ float dot8(const bf16x8& a, const float (&b)[8]) {
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < 8; ++c) s += (float)a[c] * b[c];
    return s;
}

#define LOG2E 1.4426950408889634f
#define LN2 0.6931471805599453f

// grid (ceil(W / 64), H, B * nH), block 256 = 64 pixels x 4 lanes
template <int K>
__global__ __launch_bounds__(256) void na2d_fwd_kernel(Na2d p) {
    constexpr int RB = 2 * K - 1;
    __shared__ float rpb[RB * RB];
    const int h = blockIdx.z % p.nH, b = blockIdx.z / p.nH, y = blockIdx.y;
    for (int i = threadIdx.x; i < RB * RB; i += 256) rpb[i] = p.rpb ? p.rpb[h * RB * RB + i] * LOG2E : 0.f;
    __syncthreads();
    const int c8 = (threadIdx.x & 3) * 8;
    const int xr = blockIdx.x * 64 + (threadIdx.x >> 2);
    const bool valid = xr < p.W;
    const int x = min(xr, p.W - 1);
    const int C = p.nH * 32;
    const long pix = ((long)b * p.H + y) * p.W + x;
    float q[8];
    {
        const bf16x8 qv = *(const bf16x8*)(p.qkv + pix * 3 * C + h * 32 + c8);
#pragma unroll
        for (int c = 0; c < 8; ++c) q[c] = (float)qv[c] * (p.scale * LOG2E);
    }
    const AxisWin wy = axis_win(y, p.H, p.d, K), wx = axis_win(x, p.W, p.d, K);
    float m = -INFINITY, l = 0.f, acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int i = 0; i < K; ++i) {
        const int ky = (wy.start + i) * p.d + wy.r;
        const bf16* row = p.qkv + (((long)b * p.H + ky) * p.W) * 3 * C + C + h * 32 + c8;
        float s[K];
        bf16x8 vv[K];
        float mx = m;
#pragma unroll
        for (int j = 0; j < K; ++j) {
            const bf16* kp = row + (long)((wx.start + j) * p.d + wx.r) * 3 * C;
            const bf16x8 kv = *(const bf16x8*)kp;
            vv[j] = *(const bf16x8*)(kp + C);
            s[j] = quad_sum(dot8(kv, q)) + rpb[(wy.pb0 + i) * RB + wx.pb0 + j];
            mx = fmaxf(mx, s[j]);
        }
        const float corr = fast_exp2(m - mx);          // first row: exp2(-inf) = 0
        l *= corr;
#pragma unroll
        for (int c = 0; c < 8; ++c) acc[c] *= corr;
#pragma unroll
        for (int j = 0; j < K; ++j) {
            const float pj = fast_exp2(s[j] - mx);
            l += pj;
#pragma unroll
            for (int c = 0; c < 8; ++c) acc[c] += pj * (float)vv[j][c];
        }
        m = mx;
    }
    if (!valid) return;
    const float inv = 1.0f / l;
    bf16x8 o;
#pragma unroll
    for (int c = 0; c < 8; ++c) o[c] = (bf16)(acc[c] * inv);
    *(bf16x8*)(p.out + pix * C + h * 32 + c8) = o;
    if (c8 == 0 && p.lse) p.lse[(((long)b * p.nH + h) * p.H + y) * p.W + x] = (m + __log2f(l)) * LN2;
}

// dq, drpb and delta = dout . out.  Same grid; drpb partial sums per block in LDS (one LDS atomic per wave and window slot
// when the wave's 16 pixels share the bias entry -- always, away from the left / right border), then one global atomic per bin.
template <int K>
__global__ __launch_bounds__(256) void na2d_bwd_q_kernel(Na2d p) {
    constexpr int RB = 2 * K - 1;
    __shared__ float rpb[RB * RB];
    __shared__ float dbin[RB * RB];
    const int h = blockIdx.z % p.nH, b = blockIdx.z / p.nH, y = blockIdx.y;
    for (int i = threadIdx.x; i < RB * RB; i += 256) { rpb[i] = p.rpb ? p.rpb[h * RB * RB + i] : 0.f; dbin[i] = 0.f; }
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int c8 = (threadIdx.x & 3) * 8;
    const int xr = blockIdx.x * 64 + (threadIdx.x >> 2);
    const bool valid = xr < p.W;
    const int x = min(xr, p.W - 1);
    const int C = p.nH * 32;
    const long pix = ((long)b * p.H + y) * p.W + x;
    float q[8], g[8];
    float delta;
    {
        const bf16x8 qv = *(const bf16x8*)(p.qkv + pix * 3 * C + h * 32 + c8);
        const bf16x8 gv = *(const bf16x8*)(p.dout + pix * C + h * 32 + c8);
        const bf16x8 ov = *(const bf16x8*)(p.out + pix * C + h * 32 + c8);
        float dl = 0.f;
#pragma unroll
        for (int c = 0; c < 8; ++c) { q[c] = (float)qv[c]; g[c] = (float)gv[c]; dl += g[c] * (float)ov[c]; }
        delta = quad_sum(dl);
    }
    const long si = (((long)b * p.nH + h) * p.H + y) * p.W + x;
    const float lse = p.lse[si];
    if (valid && c8 == 0) p.delta[si] = delta;
    const AxisWin wy = axis_win(y, p.H, p.d, K), wx = axis_win(x, p.W, p.d, K);
    // the wave's 16 pixels share the bias column iff their pb0 agree (tail lanes were clamped to a valid pixel: masked below)
    const int pb_first = __shfl(wx.pb0, 0);
    const bool uniform = __all(wx.pb0 == pb_first || !valid);
    float dq[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int i = 0; i < K; ++i) {
        const int ky = (wy.start + i) * p.d + wy.r;
        const bf16* row = p.qkv + (((long)b * p.H + ky) * p.W) * 3 * C + C + h * 32 + c8;
#pragma unroll
        for (int j = 0; j < K; ++j) {
            const bf16* kp = row + (long)((wx.start + j) * p.d + wx.r) * 3 * C;
            const bf16x8 kv = *(const bf16x8*)kp;
            const bf16x8 vv = *(const bf16x8*)(kp + C);
            const int bin = (wy.pb0 + i) * RB + wx.pb0 + j;
            const float s = quad_sum(dot8(kv, q)) * p.scale + rpb[bin];
            const float pr = __expf(s - lse);
            const float dp = quad_sum(dot8(vv, g));
            float ds = valid ? pr * (dp - delta) : 0.f;
#pragma unroll
            for (int c = 0; c < 8; ++c) dq[c] += ds * (float)kv[c];
            if (p.drpb) {
                if (uniform) {
                    ds += __shfl_xor(ds, 4); ds += __shfl_xor(ds, 8); ds += __shfl_xor(ds, 16); ds += __shfl_xor(ds, 32);
                    if (lane == 0) atomicAdd(&dbin[(wy.pb0 + i) * RB + pb_first + j], ds);
                } else if (valid && c8 == 0) {
                    atomicAdd(&dbin[bin], ds);
                }
            }
        }
    }
    if (valid) {
        bf16x8 o;
#pragma unroll
        for (int c = 0; c < 8; ++c) o[c] = (bf16)(dq[c] * p.scale);
        *(bf16x8*)(p.dqkv + pix * 3 * C + h * 32 + c8) = o;
    }
    if (p.drpb) {
        __syncthreads();
        for (int i = threadIdx.x; i < RB * RB; i += 256)
            if (dbin[i] != 0.f) atomicAdd(p.drpb + h * RB * RB + i, dbin[i]);
    }
}

// dk, dv: one pixel-head per 4 lanes as a KEY, looping over the queries whose window contains it (NATTEN's inverse
// neighbourhood: class positions [p < k ? 0 : p - k/2, p >= L - k ? L : p + k/2 + 1) per axis).
template <int K>
__global__ __launch_bounds__(256) void na2d_bwd_kv_kernel(Na2d p) {
    constexpr int RB = 2 * K - 1, NS = K / 2;
    __shared__ float rpb[RB * RB];
    const int h = blockIdx.z % p.nH, b = blockIdx.z / p.nH, y = blockIdx.y;
    for (int i = threadIdx.x; i < RB * RB; i += 256) rpb[i] = p.rpb ? p.rpb[h * RB * RB + i] : 0.f;
    __syncthreads();
    const int c8 = (threadIdx.x & 3) * 8;
    const int x = blockIdx.x * 64 + (threadIdx.x >> 2);
    if (x >= p.W) return;                          // whole quads leave together: the quad shuffles below stay well defined
    const int C = p.nH * 32;
    const long pix = ((long)b * p.H + y) * p.W + x;
    float kf[8], vf[8];
    {
        const bf16x8 kv = *(const bf16x8*)(p.qkv + pix * 3 * C + C + h * 32 + c8);
        const bf16x8 vv = *(const bf16x8*)(p.qkv + pix * 3 * C + 2 * C + h * 32 + c8);
#pragma unroll
        for (int c = 0; c < 8; ++c) { kf[c] = (float)kv[c]; vf[c] = (float)vv[c]; }
    }
    const int ry = y % p.d, py = y / p.d, Ly = (p.H - ry + p.d - 1) / p.d;
    const int rx = x % p.d, px = x / p.d, Lx = (p.W - rx + p.d - 1) / p.d;
    const int qy0 = py < K ? 0 : py - NS, qy1 = py >= Ly - K ? Ly : py + NS + 1;
    const int qx0 = px < K ? 0 : px - NS, qx1 = px >= Lx - K ? Lx : px + NS + 1;
    float dk[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, dv[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int qpy = qy0; qpy < qy1; ++qpy) {
        const int qy = qpy * p.d + ry;
        const int by = py - qpy + K - 1;                                  // bias row of this key in that query's window
        const long rowpix = ((long)b * p.H + qy) * p.W;
        const long rowsi = (((long)b * p.nH + h) * p.H + qy) * p.W;
        for (int qpx = qx0; qpx < qx1; ++qpx) {
            const int qx = qpx * p.d + rx;
            const bf16x8 qv = *(const bf16x8*)(p.qkv + (rowpix + qx) * 3 * C + h * 32 + c8);
            const bf16x8 gv = *(const bf16x8*)(p.dout + (rowpix + qx) * C + h * 32 + c8);
            const float lse = p.lse[rowsi + qx], delta = p.delta[rowsi + qx];
            const float s = quad_sum(dot8(qv, kf)) * p.scale + rpb[by * RB + px - qpx + K - 1];
            const float pr = __expf(s - lse);
            const float dp = quad_sum(dot8(gv, vf));
            const float ds = pr * (dp - delta);
#pragma unroll
            for (int c = 0; c < 8; ++c) { dv[c] += pr * (float)gv[c]; dk[c] += ds * (float)qv[c]; }
        }
    }
    bf16x8 ok, ov;
#pragma unroll
    for (int c = 0; c < 8; ++c) { ok[c] = (bf16)(dk[c] * p.scale); ov[c] = (bf16)dv[c]; }
    *(bf16x8*)(p.dqkv + pix * 3 * C + C + h * 32 + c8) = ok;
    *(bf16x8*)(p.dqkv + pix * 3 * C + 2 * C + h * 32 + c8) = ov;
}

static int na2d_check(const Na2d& p, int K) {
    UENC_CHECK_ARG(p.B > 0 && p.H > 0 && p.W > 0 && p.nH > 0 && p.d >= 1);
    UENC_CHECK_ARG(K >= 3 && K <= 13 && (K & 1));
    UENC_CHECK_ARG(p.H >= K * p.d && p.W >= K * p.d);      // the caller zero-pads smaller inputs first, like NATTEN
    UENC_CHECK_ARG((long)p.B * p.nH <= 65535 && p.H <= 65535);
    return UENC_OK;
}

#define NA2D_DISPATCH(KERNEL)                                                                                   \
    switch (K) {                                                                                                \
        case 3: hipLaunchKernelGGL(KERNEL<3>, grid, dim3(256), 0, stream, p); break;                            \
        case 5: hipLaunchKernelGGL(KERNEL<5>, grid, dim3(256), 0, stream, p); break;                            \
        case 7: hipLaunchKernelGGL(KERNEL<7>, grid, dim3(256), 0, stream, p); break;                            \
        case 9: hipLaunchKernelGGL(KERNEL<9>, grid, dim3(256), 0, stream, p); break;                            \
        case 11: hipLaunchKernelGGL(KERNEL<11>, grid, dim3(256), 0, stream, p); break;                          \
        default: hipLaunchKernelGGL(KERNEL<13>, grid, dim3(256), 0, stream, p); break;                          \
    }

extern "C" int uenc_na2d_fwd(const void* qkv, const float* rpb, void* out, float* lse, int B, int H, int W, int nH, int K, int dilation,
                             float scale, hipStream_t stream) {
    UENC_CHECK_ARG(qkv && out);
    UENC_CHECK_ARG(((uintptr_t)qkv & 15) == 0 && ((uintptr_t)out & 15) == 0);
    Na2d p = {};
    p.qkv = (const bf16*)qkv; p.rpb = rpb; p.out = (bf16*)out; p.lse = lse;
    p.B = B; p.H = H; p.W = W; p.nH = nH; p.d = dilation; p.scale = scale;
    const int rc = na2d_check(p, K);
    if (rc != UENC_OK) return rc;
    const dim3 grid((W + 63) / 64, H, B * nH);
    NA2D_DISPATCH(na2d_fwd_kernel);
    UENC_LAUNCH_RET();
}

// dqkv (B, H, W, 3, nH, 32) bf16 is written completely; drpb (nH, 2K-1, 2K-1) fp32 is ACCUMULATED into (may be null);
// delta_ws: B * nH * H * W floats of scratch.
extern "C" int uenc_na2d_bwd(const void* qkv, const float* rpb, const void* out, const void* dout, const float* lse, void* dqkv, float* drpb,
                             float* delta_ws, int B, int H, int W, int nH, int K, int dilation, float scale, hipStream_t stream) {
    UENC_CHECK_ARG(qkv && out && dout && lse && dqkv && delta_ws);
    UENC_CHECK_ARG((((uintptr_t)qkv | (uintptr_t)out | (uintptr_t)dout | (uintptr_t)dqkv) & 15) == 0);
    Na2d p = {};
    p.qkv = (const bf16*)qkv; p.rpb = rpb; p.out = (bf16*)out; p.lse = (float*)lse; p.dout = (const bf16*)dout;
    p.dqkv = (bf16*)dqkv; p.drpb = drpb; p.delta = delta_ws;
    p.B = B; p.H = H; p.W = W; p.nH = nH; p.d = dilation; p.scale = scale;
    const int rc = na2d_check(p, K);
    if (rc != UENC_OK) return rc;
    const dim3 grid((W + 63) / 64, H, B * nH);
    NA2D_DISPATCH(na2d_bwd_q_kernel);
    NA2D_DISPATCH(na2d_bwd_kv_kernel);
    UENC_LAUNCH_RET();
}
