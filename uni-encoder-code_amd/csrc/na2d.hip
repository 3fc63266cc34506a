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
#include "na2d.h"

// Cross-lane sums on DPP (VALU data-parallel primitives): __shfl_xor compiles to ds_bpermute, which goes through the LDS
// crossbar and competes with the halo reads of the tiled kernels.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_add(float v) {
    return v + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xf, true));
}
__device__ __forceinline__ float quad_sum(float v) {       // over the 4 lanes of a pixel: quad_perm [1,0,3,2], then [2,3,0,1]
    v = dpp_add<0xB1, 0xf>(v);
    v = dpp_add<0x4E, 0xf>(v);
    return v;
}
// sum over the wave's 16 pixels of a value that is equal in the 4 lanes of each pixel; valid in lane 63 only:
// row_shr:4, row_shr:8 (lanes 12..15 of each 16-lane row hold the row's sum), row_bcast:15 into rows 1 / 3, row_bcast:31 into rows 2 / 3
__device__ __forceinline__ float pixels_sum_lane63(float v) {
    v = dpp_add<0x114, 0xf>(v);
    v = dpp_add<0x118, 0xf>(v);
    v = dpp_add<0x142, 0xa>(v);
    v = dpp_add<0x143, 0xc>(v);
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

// ---- LDS-tiled forward ----
// One workgroup = one head of one image, one residue class (y mod d, x mod d), a tile of 8 x 16 class positions.  The k and v
// rows of the tile's halo (the union of its windows: at most 14 x 22 class positions, 39 KB as bf16) are staged in LDS once --
// 2.4x the tile's own pixels instead of the 49 window slots per pixel the direct kernel pulls through L1 -- and the k x k loop
// reads them with conflict-free 16-byte LDS loads (a wave = 16 adjacent pixels = 1 KB contiguous).  q.k runs on v_dot2_f32_bf16.
typedef __bf16 bf16x2v __attribute__((ext_vector_type(2)));

__device__ __forceinline__ float dot8_bf16(const bf16x8& a, const bf16x8& b) {
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < 8; c += 2) {
        bf16x2v x, y;
        x[0] = a[c]; x[1] = a[c + 1]; y[0] = b[c]; y[1] = b[c + 1];
        s = __builtin_amdgcn_fdot2_f32_bf16(x, y, s, false);
    }
    return s;
}

#define NA_TH 8
#define NA_TW 16

// grid (tiles_x * d, tiles_y * d, B * nH), block 512 = 128 pixels x 4 lanes
template <int K>
__global__ __launch_bounds__(512) void na2d_fwd_tiled_kernel(Na2d p) {
    constexpr int RB = 2 * K - 1, NS = K / 2, HH = NA_TH + K - 1, HW = NA_TW + K - 1;
    __shared__ __attribute__((aligned(16))) bf16 ks[HH * HW * 32];
    __shared__ __attribute__((aligned(16))) bf16 vs[HH * HW * 32];
    __shared__ float rpb[RB * RB];
    const int h = blockIdx.z % p.nH, b = blockIdx.z / p.nH;
    const int ry = blockIdx.y % p.d, rx = blockIdx.x % p.d;
    const int py0 = (blockIdx.y / p.d) * NA_TH, px0 = (blockIdx.x / p.d) * NA_TW;
    const int Ly = (p.H - ry + p.d - 1) / p.d, Lx = (p.W - rx + p.d - 1) / p.d;
    if (py0 >= Ly || px0 >= Lx) return;                    // (uniform: before any barrier)
    const int C = p.nH * 32;
    // halo = [hsy, hey) x [hsx, hex) in class positions: from the first pixel's window start to the last pixel's window end
    const int hsy = min(max(py0 - NS, 0), Ly - K), hey = min(max(min(py0 + NA_TH, Ly) - 1 - NS, 0), Ly - K) + K;
    const int hsx = min(max(px0 - NS, 0), Lx - K), hex = min(max(min(px0 + NA_TW, Lx) - 1 - NS, 0), Lx - K) + K;
    const int hh = hey - hsy, hw = hex - hsx;
    for (int i = threadIdx.x; i < RB * RB; i += 512) rpb[i] = p.rpb ? p.rpb[h * RB * RB + i] * LOG2E : 0.f;
    for (int i = threadIdx.x; i < hh * hw * 4; i += 512) {
        const int c = (i & 3) * 8, px = i >> 2;
        const int yy = px / hw, xx = px - yy * hw;
        const bf16* src = p.qkv + (((long)b * p.H + (hsy + yy) * p.d + ry) * p.W + (hsx + xx) * p.d + rx) * 3 * C + C + h * 32 + c;
        *(bf16x8*)(ks + (yy * HW + xx) * 32 + c) = *(const bf16x8*)src;
        *(bf16x8*)(vs + (yy * HW + xx) * 32 + c) = *(const bf16x8*)(src + C);
    }
    __syncthreads();
    const int c8 = (threadIdx.x & 3) * 8;
    const int ql = threadIdx.x >> 2;
    const int pyr = py0 + ql / NA_TW, pxr = px0 + ql % NA_TW;
    const bool valid = pyr < Ly && pxr < Lx;
    const int py = min(pyr, Ly - 1), px = min(pxr, Lx - 1);
    const int y = py * p.d + ry, x = px * p.d + rx;
    const long pix = ((long)b * p.H + y) * p.W + x;
    const bf16x8 q = *(const bf16x8*)(p.qkv + pix * 3 * C + h * 32 + c8);
    const int sy = min(max(py - NS, 0), Ly - K), sx = min(max(px - NS, 0), Lx - K);
    const int pby = sy - py + K - 1, pbx = sx - px + K - 1;
    const bf16* kb = ks + ((sy - hsy) * HW + (sx - hsx)) * 32 + c8;
    const bf16* vb = vs + ((sy - hsy) * HW + (sx - hsx)) * 32 + c8;
    const float qs = p.scale * LOG2E;
    float m = -INFINITY, l = 0.f, acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int i = 0; i < K; ++i) {
        float s[K];
        float mx = m;
#pragma unroll
        for (int j = 0; j < K; ++j) {
            const bf16x8 kv = *(const bf16x8*)(kb + (i * HW + j) * 32);
            s[j] = quad_sum(dot8_bf16(kv, q)) * qs + rpb[(pby + i) * RB + pbx + j];
            mx = fmaxf(mx, s[j]);
        }
        const float corr = fast_exp2(m - mx);
        l *= corr;
#pragma unroll
        for (int c = 0; c < 8; ++c) acc[c] *= corr;
#pragma unroll
        for (int j = 0; j < K; ++j) {
            const bf16x8 vv = *(const bf16x8*)(vb + (i * HW + j) * 32);
            const float pj = fast_exp2(s[j] - mx);
            l += pj;
#pragma unroll
            for (int c = 0; c < 8; ++c) acc[c] += pj * (float)vv[c];
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
                    ds = pixels_sum_lane63(ds);
                    if (lane == 63) atomicAdd(&dbin[(wy.pb0 + i) * RB + pb_first + j], ds);
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

// ---- LDS-tiled backward, per query: dq, drpb, delta (same tile / halo as the forward) ----
template <int K>
__global__ __launch_bounds__(512) void na2d_bwd_q_tiled_kernel(Na2d p) {
    constexpr int RB = 2 * K - 1, NS = K / 2, HH = NA_TH + K - 1, HW = NA_TW + K - 1;
    __shared__ __attribute__((aligned(16))) bf16 ks[HH * HW * 32];
    __shared__ __attribute__((aligned(16))) bf16 vs[HH * HW * 32];
    __shared__ float rpb[RB * RB];
    __shared__ float dbin[RB * RB];
    const int h = blockIdx.z % p.nH, b = blockIdx.z / p.nH;
    const int ry = blockIdx.y % p.d, rx = blockIdx.x % p.d;
    const int py0 = (blockIdx.y / p.d) * NA_TH, px0 = (blockIdx.x / p.d) * NA_TW;
    const int Ly = (p.H - ry + p.d - 1) / p.d, Lx = (p.W - rx + p.d - 1) / p.d;
    if (py0 >= Ly || px0 >= Lx) return;
    const int C = p.nH * 32;
    const int hsy = min(max(py0 - NS, 0), Ly - K), hey = min(max(min(py0 + NA_TH, Ly) - 1 - NS, 0), Ly - K) + K;
    const int hsx = min(max(px0 - NS, 0), Lx - K), hex = min(max(min(px0 + NA_TW, Lx) - 1 - NS, 0), Lx - K) + K;
    const int hh = hey - hsy, hw = hex - hsx;
    for (int i = threadIdx.x; i < RB * RB; i += 512) { rpb[i] = p.rpb ? p.rpb[h * RB * RB + i] * LOG2E : 0.f; dbin[i] = 0.f; }
    for (int i = threadIdx.x; i < hh * hw * 4; i += 512) {
        const int c = (i & 3) * 8, px = i >> 2;
        const int yy = px / hw, xx = px - yy * hw;
        const bf16* src = p.qkv + (((long)b * p.H + (hsy + yy) * p.d + ry) * p.W + (hsx + xx) * p.d + rx) * 3 * C + C + h * 32 + c;
        *(bf16x8*)(ks + (yy * HW + xx) * 32 + c) = *(const bf16x8*)src;
        *(bf16x8*)(vs + (yy * HW + xx) * 32 + c) = *(const bf16x8*)(src + C);
    }
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int c8 = (threadIdx.x & 3) * 8;
    const int ql = threadIdx.x >> 2;
    const int pyr = py0 + ql / NA_TW, pxr = px0 + ql % NA_TW;
    const bool valid = pyr < Ly && pxr < Lx;
    const int py = min(pyr, Ly - 1), px = min(pxr, Lx - 1);
    const int y = py * p.d + ry, x = px * p.d + rx;
    const long pix = ((long)b * p.H + y) * p.W + x;
    const bf16x8 q = *(const bf16x8*)(p.qkv + pix * 3 * C + h * 32 + c8);
    const bf16x8 g = *(const bf16x8*)(p.dout + pix * C + h * 32 + c8);
    float delta;
    {
        const bf16x8 ov = *(const bf16x8*)(p.out + pix * C + h * 32 + c8);
        float dl = 0.f;
#pragma unroll
        for (int c = 0; c < 8; ++c) dl += (float)g[c] * (float)ov[c];
        delta = quad_sum(dl);
    }
    const long si = (((long)b * p.nH + h) * p.H + y) * p.W + x;
    const float lse2 = p.lse[si] * LOG2E;
    if (valid && c8 == 0) p.delta[si] = delta;
    const int sy = min(max(py - NS, 0), Ly - K), sx = min(max(px - NS, 0), Lx - K);
    const int pby = sy - py + K - 1, pbx = sx - px + K - 1;
    const bf16* kb = ks + ((sy - hsy) * HW + (sx - hsx)) * 32 + c8;
    const bf16* vb = vs + ((sy - hsy) * HW + (sx - hsx)) * 32 + c8;
    const float qs = p.scale * LOG2E;
    // a wave = the 16 pixels of one tile row: they share the bias row, and the bias column too unless the row touches a border
    const int pb_first = __shfl(pbx, 0);
    const bool uniform = __all(pbx == pb_first || !valid);
    float dq[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int i = 0; i < K; ++i) {
#pragma unroll
        for (int j = 0; j < K; ++j) {
            const bf16x8 kv = *(const bf16x8*)(kb + (i * HW + j) * 32);
            const bf16x8 vv = *(const bf16x8*)(vb + (i * HW + j) * 32);
            const int bin = (pby + i) * RB + pbx + j;
            const float s = quad_sum(dot8_bf16(kv, q)) * qs + rpb[bin];
            const float pr = fast_exp2(s - lse2);
            const float dp = quad_sum(dot8_bf16(vv, g));
            float ds = valid ? pr * (dp - delta) : 0.f;
#pragma unroll
            for (int c = 0; c < 8; ++c) dq[c] += ds * (float)kv[c];
            if (p.drpb) {
                if (uniform) {
                    ds = pixels_sum_lane63(ds);
                    if (lane == 63) atomicAdd(&dbin[(pby + i) * RB + pb_first + j], ds);
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
        for (int i = threadIdx.x; i < RB * RB; i += 512)
            if (dbin[i] != 0.f) atomicAdd(p.drpb + h * RB * RB + i, dbin[i]);
    }
}

// ---- LDS-tiled backward, per key: dk, dv.  Tile of 8 x 16 KEY class positions; the q and dout rows (and lse, delta) of every
// query whose window can contain one of them -- NATTEN's inverse neighbourhood: T + K - 1 positions per axis in the interior,
// T + K + K/2 - 1 next to a border, the whole class when it is shorter than T + 2K - 1 -- are staged in LDS (58 KB at K = 7: two
// workgroups per CU).  Used for K <= 7; larger kernels use the direct form. ----
template <int K>
__global__ __launch_bounds__(512) void na2d_bwd_kv_tiled_kernel(Na2d p) {
    constexpr int RB = 2 * K - 1, NS = K / 2;
    const int HW = p.hw_max, HP = p.hh_max * p.hw_max;       // halo row stride / positions (launch-wide maxima, see the launcher)
    extern __shared__ __attribute__((aligned(16))) unsigned char dyn[];
    bf16* qs_ = (bf16*)dyn;                                  // [HP][32]
    bf16* gs = qs_ + HP * 32;                                // [HP][32]
    float* ls = (float*)(gs + HP * 32);                      // [HP] lse (log2 domain)
    float* dl = ls + HP;                                     // [HP] delta
    float* rpb = dl + HP;                                    // [RB * RB]
    const int h = blockIdx.z % p.nH, b = blockIdx.z / p.nH;
    const int ry = blockIdx.y % p.d, rx = blockIdx.x % p.d;
    const int py0 = (blockIdx.y / p.d) * NA_TH, px0 = (blockIdx.x / p.d) * NA_TW;
    const int Ly = (p.H - ry + p.d - 1) / p.d, Lx = (p.W - rx + p.d - 1) / p.d;
    if (py0 >= Ly || px0 >= Lx) return;
    const int C = p.nH * 32;
    const int pyl = min(py0 + NA_TH, Ly) - 1, pxl = min(px0 + NA_TW, Lx) - 1;          // last key of the tile
    const int hsy = py0 < K ? 0 : py0 - NS, hey = pyl >= Ly - K ? Ly : pyl + NS + 1;
    const int hsx = px0 < K ? 0 : px0 - NS, hex = pxl >= Lx - K ? Lx : pxl + NS + 1;
    const int hh = hey - hsy, hw = hex - hsx;
    for (int i = threadIdx.x; i < RB * RB; i += 512) rpb[i] = p.rpb ? p.rpb[h * RB * RB + i] * LOG2E : 0.f;
    for (int i = threadIdx.x; i < hh * hw * 4; i += 512) {
        const int c = (i & 3) * 8, px = i >> 2;
        const int yy = px / hw, xx = px - yy * hw;
        const int gy = (hsy + yy) * p.d + ry, gx = (hsx + xx) * p.d + rx;
        const long gp = ((long)b * p.H + gy) * p.W + gx;
        *(bf16x8*)(qs_ + (yy * HW + xx) * 32 + c) = *(const bf16x8*)(p.qkv + gp * 3 * C + h * 32 + c);
        *(bf16x8*)(gs + (yy * HW + xx) * 32 + c) = *(const bf16x8*)(p.dout + gp * C + h * 32 + c);
        if (c == 0) {
            const long si = (((long)b * p.nH + h) * p.H + gy) * p.W + gx;
            ls[yy * HW + xx] = p.lse[si] * LOG2E;
            dl[yy * HW + xx] = p.delta[si];
        }
    }
    __syncthreads();
    const int c8 = (threadIdx.x & 3) * 8;
    const int ql = threadIdx.x >> 2;
    const int py = py0 + ql / NA_TW, px = px0 + ql % NA_TW;
    if (py >= Ly || px >= Lx) return;                        // whole quads leave together
    const int y = py * p.d + ry, x = px * p.d + rx;
    const long pix = ((long)b * p.H + y) * p.W + x;
    const bf16x8 kv = *(const bf16x8*)(p.qkv + pix * 3 * C + C + h * 32 + c8);
    const bf16x8 vv = *(const bf16x8*)(p.qkv + pix * 3 * C + 2 * C + h * 32 + c8);
    const int qy0 = py < K ? 0 : py - NS, qy1 = py >= Ly - K ? Ly : py + NS + 1;
    const int qx0 = px < K ? 0 : px - NS, qx1 = px >= Lx - K ? Lx : px + NS + 1;
    const float qsc = p.scale * LOG2E;
    float dk[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, dv[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int qpy = qy0; qpy < qy1; ++qpy) {
        const int by = py - qpy + K - 1;
        const int rowl = (qpy - hsy) * HW - hsx;
        for (int qpx = qx0; qpx < qx1; ++qpx) {
            const int li = rowl + qpx;
            const bf16x8 qv = *(const bf16x8*)(qs_ + li * 32 + c8);
            const bf16x8 gv = *(const bf16x8*)(gs + li * 32 + c8);
            const float s = quad_sum(dot8_bf16(qv, kv)) * qsc + rpb[by * RB + px - qpx + K - 1];
            const float pr = fast_exp2(s - ls[li]);
            const float dp = quad_sum(dot8_bf16(gv, vv));
            const float ds = pr * (dp - dl[li]);
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

#define NA2D_DISPATCH512(KERNEL)                                                                                \
switch (K) {                                                                                            \
    case 3: hipLaunchKernelGGL(KERNEL<3>, grid, dim3(512), 0, stream, p); break;                        \
    case 5: hipLaunchKernelGGL(KERNEL<5>, grid, dim3(512), 0, stream, p); break;                        \
    case 7: hipLaunchKernelGGL(KERNEL<7>, grid, dim3(512), 0, stream, p); break;                        \
    case 9: hipLaunchKernelGGL(KERNEL<9>, grid, dim3(512), 0, stream, p); break;                        \
    case 11: hipLaunchKernelGGL(KERNEL<11>, grid, dim3(512), 0, stream, p); break;                      \
    default: hipLaunchKernelGGL(KERNEL<13>, grid, dim3(512), 0, stream, p); break;                      \
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
    const char* ev = getenv("UENC_NA2D_VARIANT");              // A/B: 1 = the direct (L1-gather) kernels, 2 = the LDS-tiled VALU kernels
    if (na2d_mfma_supported(H, W, nH, K, dilation) && !(ev && (atoi(ev) & 3))) return na2d_mfma_fwd(p, K, stream);          // matrix cores (na2d_mfma.hip)
    if (ev && (atoi(ev) & 1)) {
        const dim3 grid((W + 63) / 64, H, B * nH);
        NA2D_DISPATCH(na2d_fwd_kernel);
    } else {
        const int Lx = (W + dilation - 1) / dilation, Ly = (H + dilation - 1) / dilation;      // the longest residue class
        UENC_CHECK_ARG((long)((Ly + NA_TH - 1) / NA_TH) * dilation <= 65535);
        const dim3 grid(((Lx + NA_TW - 1) / NA_TW) * dilation, ((Ly + NA_TH - 1) / NA_TH) * dilation, B * nH);
        NA2D_DISPATCH512(na2d_fwd_tiled_kernel);
    }
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
    const char* ev = getenv("UENC_NA2D_VARIANT");
    if (na2d_mfma_supported(H, W, nH, K, dilation) && !(ev && (atoi(ev) & 3))) return na2d_mfma_bwd(p, K, stream);           // matrix cores (na2d_mfma.hip)
    const bool direct = ev && (atoi(ev) & 1);                 // A/B: the direct (L1-gather) kernels
    const int Lx = (W + dilation - 1) / dilation, Ly = (H + dilation - 1) / dilation;
    const dim3 tgrid(((Lx + NA_TW - 1) / NA_TW) * dilation, ((Ly + NA_TH - 1) / NA_TH) * dilation, B * nH);
    UENC_CHECK_ARG(tgrid.y <= 65535);
    {
        dim3 grid = tgrid;
        if (direct) { grid = dim3((W + 63) / 64, H, B * nH); NA2D_DISPATCH(na2d_bwd_q_kernel); }
        else { NA2D_DISPATCH512(na2d_bwd_q_tiled_kernel); }
    }
    if (direct || K > 7) {
        const dim3 grid((W + 63) / 64, H, B * nH);
        NA2D_DISPATCH(na2d_bwd_kv_kernel);
    } else {
        // halo extent per axis: T + K + K/2 - 1 beside one border; a class shorter than T + 2K - 1 can touch both: all of it
        const int NSh = K / 2;
        p.hh_max = Ly >= NA_TH + 2 * K - 1 ? NA_TH + K + NSh - 1 : (Ly < NA_TH + 2 * K - 2 ? Ly : NA_TH + 2 * K - 2);
        p.hw_max = Lx >= NA_TW + 2 * K - 1 ? NA_TW + K + NSh - 1 : (Lx < NA_TW + 2 * K - 2 ? Lx : NA_TW + 2 * K - 2);
        const int SHM = p.hh_max * p.hw_max * (2 * 64 + 8) + (2 * K - 1) * (2 * K - 1) * 4;
#define NA2D_KV_TILED(KK)                                                                                                           \
        {                                                                                                                            \
            static int attr = 0;                                                                                                     \
            if (attr < SHM) {                                                                                                        \
                hipError_t e = hipFuncSetAttribute((const void*)na2d_bwd_kv_tiled_kernel<KK>, hipFuncAttributeMaxDynamicSharedMemorySize, SHM); \
                if (e != hipSuccess) return (int)e;                                                                                  \
                attr = SHM;                                                                                                          \
            }                                                                                                                        \
            hipLaunchKernelGGL(na2d_bwd_kv_tiled_kernel<KK>, tgrid, dim3(512), SHM, stream, p);                                      \
        }
        if (K == 3) NA2D_KV_TILED(3) else if (K == 5) NA2D_KV_TILED(5) else NA2D_KV_TILED(7)
    }
    UENC_LAUNCH_RET();
}
