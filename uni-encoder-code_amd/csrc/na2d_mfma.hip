// Neighbourhood attention 2-D on the matrix cores (window sizes 3, 5, 7; head_dim 32) -- the DiNAT hot kernels.
// Same arithmetic and layouts as na2d.hip (see its header); this file replaces the per-lane dot products by MFMA tiles.
//
// Tiling.  A workgroup (8 waves) owns one head of one image, one residue class (y mod d, x mod d) and 8 x 16 class positions; the k / v
// rows of the tile's halo (<= 14 x 22 positions at K = 7) are staged once in LDS as [position][32] bf16 images (64-byte rows, the XOR
// swizzle of lds_frag.h).  A wave owns a 2 x 8 block of queries: the union of their windows is at most (K + 1) x (K + 7) <= 8 x 14
// positions, i.e. K + 1 key blocks of 16 consecutive halo positions (one halo row each, the columns past the window masked).  Per key
// block one v_mfma_f32_16x16x32_bf16 gives S^T = K Q^T (head_dim 32 = the instruction's whole k extent): a lane owns ONE query column
// and 4 keys per block, so the softmax statistics are lane-local plus two cross-lane steps, and P^T leaves the accumulators straight into
// the PV product as its B operand (k-slots in accumulator order, V^T through the transposing LDS read) -- P never touches LDS.
// 2 (K + 1) MFMAs per 16 queries at 49 / 128 useful scores (K = 7), against 49 x 26 VALU instructions per query before.
//
// Bias: rpb[key - query + K - 1] per axis.  The 4 scores a lane holds per block are 4 consecutive bias columns: the table sits in LDS
// in 4 copies shifted by 0..3 floats, so that every lane reads its 4 values with one aligned ds_read_b128.
//
// Backward, per query (dq, drpb, delta): the same tiles plus dP^T = V dO^T, dS^T = P^T (dP^T - delta), dQ^T += K^T dS^T.  drpb is a
// sum of dS over the queries along diagonals (bin = key - query): each wave lays its dS block out in LDS as [query][key column] and 46
// lanes add along the diagonals.  Backward, per key (dk, dv): a wave owns 2 x 8 KEYS and sweeps the blocks of 16 queries of their
// inverse neighbourhood (S = Q K^T orientation: a lane owns one key column), dV^T += dO^T P, dK^T += Q^T dS, no atomics.
#include "common.h"
#include "lds_frag.h"
#include "na2d.h"

#define LOG2E 1.4426950408889634f
#define LN2 0.6931471805599453f
#define NM_TH 8
#define NM_TW 16
#define NEG_BIG (-1e30f)

template <int K>
struct NmGeom {
    static constexpr int NS = K / 2, RB = 2 * K - 1, NKT = K + 1;
    static constexpr int HH = NM_TH + K;                                 // halo rows + one spare (a wave's last key block may be all padding)
    static constexpr int HWP = ((NM_TW + K - 1 + 7) / 8) * 8;            // halo row stride in positions
    static constexpr int NPOS = HH * HWP + 8;                            // + tail: a 16-position block starts at column <= 15 of a 24-wide row
    static constexpr int TAB = 4 * RB * 32;                              // bias table: 4 shifted copies of [RB][32]
};

__device__ __forceinline__ int clampi(int v, int lo, int hi) { return min(max(v, lo), hi); }

// stage the halo [hsy, hsy + hh) x [hsx, hsx + hw) of one head slice (element offset `off` inside a pixel's `stride` elements) as an
// LDS image of NPOS positions; positions outside the halo are zero (they are read as masked operands: must be finite)
template <int NPOS, int HWP>
__device__ __forceinline__ void nm_stage(unsigned char* img, const bf16* src, long stride, const Na2d& p, int b, int ry, int rx,
                                         int hsy, int hsx, int hh, int hw) {
    for (int i = threadIdx.x; i < NPOS * 4; i += 512) {
        const int c = i & 3, pos = i >> 2;
        const int yy = pos / HWP, xx = pos - yy * HWP;
        u32x4 v = {0u, 0u, 0u, 0u};
        if (yy < hh && xx < hw)
            v = *(const u32x4*)(src + (((long)b * p.H + (hsy + yy) * p.d + ry) * p.W + (hsx + xx) * p.d + rx) * stride + c * 8);
        *(u32x4*)(img + rm_off(pos, c)) = v;
    }
}

template <int K>
__device__ __forceinline__ void nm_bias_table(float* tab, const float* rpb, int h) {
    constexpr int RB = 2 * K - 1;
    for (int i = threadIdx.x; i < 4 * RB * 32; i += 512) {
        const int e = i / (RB * 32), rem = i - e * RB * 32, row = rem >> 5, col = (rem & 31) + e - 8;
        tab[i] = (rpb != nullptr && col >= 0 && col < RB) ? rpb[(h * RB + row) * RB + col] * LOG2E : 0.f;
    }
}

// per-lane description of a wave's 2 x 8 query block inside the workgroup's halo
struct NmQuery {
    bool wave_active, valid;
    long pix;                 // this lane's query pixel (clamped to a valid one)
    int r0;                   // image position of key block 0, key column 0
    int dy0, dx0;             // this query's window start relative to the wave's key region
    int by0, j0;              // bias row of key block 0; bias column of key column 4 fg
    long si;                  // index into (B, nH, H, W) statistics
};

template <int K>
__device__ __forceinline__ NmQuery nm_query(const Na2d& p, int b, int h, int ry, int rx, int py0, int px0, int Ly, int Lx, int hsy, int hsx) {
    constexpr int NS = K / 2, HWP = NmGeom<K>::HWP;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, fr = lane & 15, fg = lane >> 4;
    const int wy = wave >> 1, wx = wave & 1;
    NmQuery q;
    q.wave_active = (py0 + 2 * wy < Ly) && (px0 + 8 * wx < Lx);
    const int py_a = min(py0 + 2 * wy, Ly - 1), px_a = min(px0 + 8 * wx, Lx - 1);
    const int pyr = py0 + 2 * wy + (fr >> 3), pxr = px0 + 8 * wx + (fr & 7);
    q.valid = pyr < Ly && pxr < Lx;
    const int py = min(pyr, Ly - 1), px = min(pxr, Lx - 1);
    const int sy_a = clampi(py_a - NS, 0, Ly - K), sx_a = clampi(px_a - NS, 0, Lx - K);
    const int sy = clampi(py - NS, 0, Ly - K), sx = clampi(px - NS, 0, Lx - K);
    q.dy0 = sy - sy_a; q.dx0 = sx - sx_a;
    q.r0 = (sy_a - hsy) * HWP + (sx_a - hsx);
    q.by0 = sy_a - py + K - 1;
    q.j0 = 4 * fg + sx_a - px + K - 1;
    const int y = py * p.d + ry, x = px * p.d + rx;
    q.pix = ((long)b * p.H + y) * p.W + x;
    q.si = (((long)b * p.nH + h) * p.H + y) * p.W + x;
    return q;
}

__device__ __forceinline__ f32x4 nm_bias4(const float* tab, int RB, int by, int j0) {
    const int e = (j0 + 8) & 3, pos = (j0 + 8) - e;
    return *(const f32x4*)(tab + (e * RB + by) * 32 + pos);
}

__device__ __forceinline__ void nm_store4(bf16* dst, const f32x4& v, float mul) {
    bf16x4 o;
#pragma unroll
    for (int r = 0; r < 4; ++r) o[r] = (bf16)(v[r] * mul);
    *(bf16x4*)dst = o;
}

// ---------------------------------------------------------------------------------------------------------------------
// forward.  grid (tiles_x * d, tiles_y * d, B * nH), block 512
// ---------------------------------------------------------------------------------------------------------------------
template <int K>
__global__ __launch_bounds__(512) void na2d_mfma_fwd_kernel(Na2d p) {
    typedef NmGeom<K> G;
    constexpr int NS = G::NS, RB = G::RB, NKT = G::NKT, HWP = G::HWP;
    __shared__ __attribute__((aligned(16))) unsigned char ks[G::NPOS * 64];
    __shared__ __attribute__((aligned(16))) unsigned char vs[G::NPOS * 64];
    __shared__ __attribute__((aligned(16))) float tab[G::TAB];
    const int h = blockIdx.z % p.nH, b = blockIdx.z / p.nH;
    const int ry = blockIdx.y % p.d, rx = blockIdx.x % p.d;
    const int py0 = (blockIdx.y / p.d) * NM_TH, px0 = (blockIdx.x / p.d) * NM_TW;
    const int Ly = (p.H - ry + p.d - 1) / p.d, Lx = (p.W - rx + p.d - 1) / p.d;
    if (py0 >= Ly || px0 >= Lx) return;                    // (uniform: before any barrier)
    const int C = p.nH * 32;
    const int hsy = clampi(py0 - NS, 0, Ly - K), hey = clampi(min(py0 + NM_TH, Ly) - 1 - NS, 0, Ly - K) + K;
    const int hsx = clampi(px0 - NS, 0, Lx - K), hex = clampi(min(px0 + NM_TW, Lx) - 1 - NS, 0, Lx - K) + K;
    nm_bias_table<K>(tab, p.rpb, h);
    nm_stage<G::NPOS, HWP>(ks, p.qkv + C + h * 32, 3L * C, p, b, ry, rx, hsy, hsx, hey - hsy, hex - hsx);
    nm_stage<G::NPOS, HWP>(vs, p.qkv + 2 * C + h * 32, 3L * C, p, b, ry, rx, hsy, hsx, hey - hsy, hex - hsx);
    __syncthreads();
    const NmQuery q = nm_query<K>(p, b, h, ry, rx, py0, px0, Ly, Lx, hsy, hsx);
    if (!q.wave_active) return;                             // whole waves: EXEC stays full for the transposing reads below
    const int lane = threadIdx.x & 63, fr = lane & 15, fg = lane >> 4;
    const bf16x8 qf = *(const bf16x8*)(p.qkv + q.pix * 3 * C + h * 32 + 8 * fg);
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    const float sc = p.scale * LOG2E;
    bool colok[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) colok[r] = (unsigned)(4 * fg + r - q.dx0) < (unsigned)K;
    f32x4 s[NKT];
    float m = NEG_BIG;
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt) {
        s[kt] = mfma16(frag_rows(ks, q.r0 + kt * HWP, fr, fg), qf, zero4);
        const f32x4 bias = nm_bias4(tab, RB, clampi(q.by0 + kt, 0, RB - 1), q.j0);
        const bool rowok = (unsigned)(kt - q.dy0) < (unsigned)K;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            s[kt][r] = (rowok && colok[r]) ? s[kt][r] * sc + bias[r] : NEG_BIG;
            m = fmaxf(m, s[kt][r]);
        }
    }
    m = fmaxf(m, __shfl_xor(m, 16));
    m = fmaxf(m, __shfl_xor(m, 32));
    float l = 0.f;
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            s[kt][r] = fast_exp2(s[kt][r] - m);
            l += s[kt][r];
        }
    l += __shfl_xor(l, 16);
    l += __shfl_xor(l, 32);
    f32x4 o[2] = {zero4, zero4};
#pragma unroll
    for (int s2 = 0; s2 < NKT / 2; ++s2) {
        bf16x8 pb;
#pragma unroll
        for (int r = 0; r < 4; ++r) { pb[r] = (bf16)s[2 * s2][r]; pb[4 + r] = (bf16)s[2 * s2 + 1][r]; }
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
            o[dt] = mfma16(frag_tr(vs, q.r0 + 2 * s2 * HWP, q.r0 + (2 * s2 + 1) * HWP, dt * 16, lane), pb, o[dt]);
    }
    if (!q.valid) return;
    const float inv = 1.0f / l;
#pragma unroll
    for (int dt = 0; dt < 2; ++dt) nm_store4(p.out + q.pix * C + h * 32 + dt * 16 + 4 * fg, o[dt], inv);
    if (fg == 0 && p.lse) p.lse[q.si] = (m + __log2f(l)) * LN2;
}

// ---------------------------------------------------------------------------------------------------------------------
// backward per query: dq, drpb, delta.  Same grid.
// ---------------------------------------------------------------------------------------------------------------------
template <int K>
__global__ __launch_bounds__(512) void na2d_mfma_bwd_q_kernel(Na2d p) {
    typedef NmGeom<K> G;
    constexpr int NS = G::NS, RB = G::RB, NKT = G::NKT, HWP = G::HWP, NBIN = 2 * (16 + 7);
    __shared__ __attribute__((aligned(16))) unsigned char ks[G::NPOS * 64];
    __shared__ __attribute__((aligned(16))) unsigned char vs[G::NPOS * 64];
    __shared__ __attribute__((aligned(16))) float tab[G::TAB];
    __shared__ __attribute__((aligned(16))) float dsbuf[8][256];           // per wave: dS of one key block as [query][key column]
    __shared__ float dbin[RB * RB];
    const int h = blockIdx.z % p.nH, b = blockIdx.z / p.nH;
    const int ry = blockIdx.y % p.d, rx = blockIdx.x % p.d;
    const int py0 = (blockIdx.y / p.d) * NM_TH, px0 = (blockIdx.x / p.d) * NM_TW;
    const int Ly = (p.H - ry + p.d - 1) / p.d, Lx = (p.W - rx + p.d - 1) / p.d;
    if (py0 >= Ly || px0 >= Lx) return;
    const int C = p.nH * 32;
    const int hsy = clampi(py0 - NS, 0, Ly - K), hey = clampi(min(py0 + NM_TH, Ly) - 1 - NS, 0, Ly - K) + K;
    const int hsx = clampi(px0 - NS, 0, Lx - K), hex = clampi(min(px0 + NM_TW, Lx) - 1 - NS, 0, Lx - K) + K;
    nm_bias_table<K>(tab, p.rpb, h);
    for (int i = threadIdx.x; i < RB * RB; i += 512) dbin[i] = 0.f;
    nm_stage<G::NPOS, HWP>(ks, p.qkv + C + h * 32, 3L * C, p, b, ry, rx, hsy, hsx, hey - hsy, hex - hsx);
    nm_stage<G::NPOS, HWP>(vs, p.qkv + 2 * C + h * 32, 3L * C, p, b, ry, rx, hsy, hsx, hey - hsy, hex - hsx);
    __syncthreads();
    const NmQuery q = nm_query<K>(p, b, h, ry, rx, py0, px0, Ly, Lx, hsy, hsx);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, fr = lane & 15, fg = lane >> 4;
    if (q.wave_active) {
        const bf16x8 qf = *(const bf16x8*)(p.qkv + q.pix * 3 * C + h * 32 + 8 * fg);
        const bf16x8 dof = *(const bf16x8*)(p.dout + q.pix * C + h * 32 + 8 * fg);
        float delta;
        {
            const bf16x8 ov = *(const bf16x8*)(p.out + q.pix * C + h * 32 + 8 * fg);
            float part = 0.f;
#pragma unroll
            for (int c = 0; c < 8; ++c) part += (float)dof[c] * (float)ov[c];
            part += __shfl_xor(part, 16);
            part += __shfl_xor(part, 32);
            delta = part;
        }
        const float lse2 = p.lse[q.si] * LOG2E;
        if (q.valid && fg == 0) p.delta[q.si] = delta;
        const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
        const float sc = p.scale * LOG2E;
        bool colok[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) colok[r] = q.valid && (unsigned)(4 * fg + r - q.dx0) < (unsigned)K;
        // drpb: lane L < 46 sums the diagonal `slot` = key column - query column + 7 of query row L / 23, per key block
        const int bq = lane / 23, bslot = lane - bq * 23;
        float* buf = dsbuf[wave];
        float bacc[NKT];
        f32x4 ds[NKT];
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt) {
            const f32x4 sv = mfma16(frag_rows(ks, q.r0 + kt * HWP, fr, fg), qf, zero4);
            const f32x4 dp = mfma16(frag_rows(vs, q.r0 + kt * HWP, fr, fg), dof, zero4);
            const f32x4 bias = nm_bias4(tab, RB, clampi(q.by0 + kt, 0, RB - 1), q.j0);
            const bool rowok = (unsigned)(kt - q.dy0) < (unsigned)K;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float pr = (rowok && colok[r]) ? fast_exp2(sv[r] * sc + bias[r] - lse2) : 0.f;
                ds[kt][r] = pr * (dp[r] - delta);
            }
            if (p.drpb) {
                *(f32x4*)(buf + fr * 16 + 4 * fg) = ds[kt];
                __builtin_amdgcn_wave_barrier();
                float a = 0.f;
#pragma unroll
                for (int qx = 0; qx < 8; ++qx) {
                    const int c = bslot - 7 + qx;
                    const float v = buf[(((bq & 1) * 8 + qx) * 16 + (c & 15))];
                    a += (lane < NBIN && (unsigned)c < 16u) ? v : 0.f;
                }
                bacc[kt] = a;
                __builtin_amdgcn_wave_barrier();
            }
        }
        f32x4 o[2] = {zero4, zero4};
#pragma unroll
        for (int s2 = 0; s2 < NKT / 2; ++s2) {
            bf16x8 pb;
#pragma unroll
            for (int r = 0; r < 4; ++r) { pb[r] = (bf16)ds[2 * s2][r]; pb[4 + r] = (bf16)ds[2 * s2 + 1][r]; }
#pragma unroll
            for (int dt = 0; dt < 2; ++dt)
                o[dt] = mfma16(frag_tr(ks, q.r0 + 2 * s2 * HWP, q.r0 + (2 * s2 + 1) * HWP, dt * 16, lane), pb, o[dt]);
        }
        if (q.valid) {
#pragma unroll
            for (int dt = 0; dt < 2; ++dt) nm_store4(p.dqkv + q.pix * 3 * C + h * 32 + dt * 16 + 4 * fg, o[dt], p.scale);
        }
        if (p.drpb && lane < NBIN) {
            // bias row of key block kt for query row bq: by0 is this wave's value for ITS first query row (lane 0), one less per row
            const int by_first = __shfl(q.by0, 0), j_first = __shfl(q.j0, 0);      // lane 0: query (row 0, column 0), fg = 0
            const int col = bslot - 7 + j_first;
#pragma unroll
            for (int kt = 0; kt < NKT; ++kt) {
                const int row = by_first - bq + kt;
                if ((unsigned)row < (unsigned)RB && (unsigned)col < (unsigned)RB && bacc[kt] != 0.f) atomicAdd(&dbin[row * RB + col], bacc[kt]);
            }
        }
    }
    if (p.drpb) {
        __syncthreads();
        for (int i = threadIdx.x; i < RB * RB; i += 512)
            if (dbin[i] != 0.f) atomicAdd(p.drpb + h * RB * RB + i, dbin[i]);
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// backward per key: dk, dv.  Tile of 8 x 16 KEY positions; q, dout, (lse, delta) of the tile's inverse neighbourhood in LDS
// (extents hh_max x hw_max from the launcher, as in na2d_bwd_kv_tiled_kernel).
// ---------------------------------------------------------------------------------------------------------------------
template <int K>
__device__ __forceinline__ int inv_start(int pos) { return pos < K ? 0 : pos - K / 2; }
template <int K>
__device__ __forceinline__ int inv_end(int pos, int L) { return pos >= L - K ? L : pos + K / 2 + 1; }

template <int K>
__global__ __launch_bounds__(512) void na2d_mfma_bwd_kv_kernel(Na2d p) {
    constexpr int RB = 2 * K - 1, NS = K / 2;
    const int HWQ = p.hw_max, NPOS = p.hh_max * p.hw_max + 32;
    extern __shared__ __attribute__((aligned(16))) unsigned char dyn[];
    unsigned char* qs = dyn;                                  // [NPOS] x 64 B
    unsigned char* gs = qs + NPOS * 64;                       // [NPOS] x 64 B
    float2* ld = (float2*)(gs + NPOS * 64);                   // [NPOS] (lse * log2e, delta)
    float* rpb = (float*)(ld + NPOS);                         // [RB * RB]
    const int h = blockIdx.z % p.nH, b = blockIdx.z / p.nH;
    const int ry = blockIdx.y % p.d, rx = blockIdx.x % p.d;
    const int py0 = (blockIdx.y / p.d) * NM_TH, px0 = (blockIdx.x / p.d) * NM_TW;
    const int Ly = (p.H - ry + p.d - 1) / p.d, Lx = (p.W - rx + p.d - 1) / p.d;
    if (py0 >= Ly || px0 >= Lx) return;
    const int C = p.nH * 32;
    const int pyl = min(py0 + NM_TH, Ly) - 1, pxl = min(px0 + NM_TW, Lx) - 1;          // last key of the tile
    const int hsy = inv_start<K>(py0), hey = inv_end<K>(pyl, Ly);
    const int hsx = inv_start<K>(px0), hex = inv_end<K>(pxl, Lx);
    const int hh = hey - hsy, hw = hex - hsx;
    for (int i = threadIdx.x; i < RB * RB; i += 512) rpb[i] = p.rpb ? p.rpb[h * RB * RB + i] * LOG2E : 0.f;
    for (int i = threadIdx.x; i < NPOS * 4; i += 512) {
        const int c = i & 3, pos = i >> 2;
        const int yy = pos / HWQ, xx = pos - yy * HWQ;
        u32x4 vq = {0u, 0u, 0u, 0u}, vg = vq;
        float2 st = {0.f, 0.f};
        if (yy < hh && xx < hw) {
            const int gy = (hsy + yy) * p.d + ry, gx = (hsx + xx) * p.d + rx;
            const long gp = ((long)b * p.H + gy) * p.W + gx;
            vq = *(const u32x4*)(p.qkv + gp * 3 * C + h * 32 + c * 8);
            vg = *(const u32x4*)(p.dout + gp * C + h * 32 + c * 8);
            if (c == 0) {
                const long si = (((long)b * p.nH + h) * p.H + gy) * p.W + gx;
                st.x = p.lse[si] * LOG2E; st.y = p.delta[si];
            }
        }
        *(u32x4*)(qs + rm_off(pos, c)) = vq;
        *(u32x4*)(gs + rm_off(pos, c)) = vg;
        if (c == 0) ld[pos] = st;
    }
    __syncthreads();
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, fr = lane & 15, fg = lane >> 4;
    const int wy = wave >> 1, wx = wave & 1;
    if (py0 + 2 * wy >= Ly || px0 + 8 * wx >= Lx) return;     // whole waves
    const int kyr = py0 + 2 * wy + (fr >> 3), kxr = px0 + 8 * wx + (fr & 7);
    const bool kvalid = kyr < Ly && kxr < Lx;
    const int ky = min(kyr, Ly - 1), kx = min(kxr, Lx - 1);
    const long pix = ((long)b * p.H + ky * p.d + ry) * p.W + kx * p.d + rx;
    const bf16x8 kf = *(const bf16x8*)(p.qkv + pix * 3 * C + C + h * 32 + 8 * fg);
    const bf16x8 vf = *(const bf16x8*)(p.qkv + pix * 3 * C + 2 * C + h * 32 + 8 * fg);
    // the wave's query region (wave-uniform): union of the inverse neighbourhoods of its keys
    const int ky_a = min(py0 + 2 * wy, Ly - 1), ky_b = min(py0 + 2 * wy + 1, Ly - 1);
    const int kx_a = min(px0 + 8 * wx, Lx - 1), kx_b = min(px0 + 8 * wx + 7, Lx - 1);
    const int qy0 = inv_start<K>(ky_a), qy1 = inv_end<K>(ky_b, Ly);
    const int qx0 = inv_start<K>(kx_a), qx1 = inv_end<K>(kx_b, Lx);
    const int ncb = (qx1 - qx0 + 15) >> 4, nblk = (qy1 - qy0) * ncb;
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    const float sc = p.scale * LOG2E;
    f32x4 dk[2] = {zero4, zero4}, dv[2] = {zero4, zero4};
    for (int j = 0; j < nblk; j += 2) {
        int r0[2];
        f32x4 pr[2], ds[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const bool real = j + u < nblk;
            const int blk = real ? j + u : 0;
            const int row = blk / ncb, cb = blk - row * ncb;
            const int qy = qy0 + row, bx = qx0 + 16 * cb;
            r0[u] = (qy - hsy) * HWQ + (bx - hsx);
            const f32x4 sv = mfma16(frag_rows(qs, r0[u], fr, fg), kf, zero4);
            const f32x4 dp = mfma16(frag_rows(gs, r0[u], fr, fg), vf, zero4);
            const int syq = clampi(qy - NS, 0, Ly - K);
            const bool rowok = real && (unsigned)(ky - syq) < (unsigned)K;
            const int brow = clampi(ky - qy + K - 1, 0, RB - 1) * RB;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int qx = bx + 4 * fg + r;
                const int sxq = clampi(qx - NS, 0, Lx - K);
                const bool ok = rowok && qx < qx1 && (unsigned)(kx - sxq) < (unsigned)K;
                const float2 st = ld[r0[u] + 4 * fg + r];
                const float bias = rpb[brow + clampi(kx - qx + K - 1, 0, RB - 1)];
                pr[u][r] = ok ? fast_exp2(sv[r] * sc + bias - st.x) : 0.f;
                ds[u][r] = pr[u][r] * (dp[r] - st.y);
            }
        }
        bf16x8 pb, db;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            pb[r] = (bf16)pr[0][r]; pb[4 + r] = (bf16)pr[1][r];
            db[r] = (bf16)ds[0][r]; db[4 + r] = (bf16)ds[1][r];
        }
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
            dv[dt] = mfma16(frag_tr(gs, r0[0], r0[1], dt * 16, lane), pb, dv[dt]);
            dk[dt] = mfma16(frag_tr(qs, r0[0], r0[1], dt * 16, lane), db, dk[dt]);
        }
    }
    if (!kvalid) return;
#pragma unroll
    for (int dt = 0; dt < 2; ++dt) {
        nm_store4(p.dqkv + pix * 3 * C + C + h * 32 + dt * 16 + 4 * fg, dk[dt], p.scale);
        nm_store4(p.dqkv + pix * 3 * C + 2 * C + h * 32 + dt * 16 + 4 * fg, dv[dt], 1.0f);
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------------------------------------------------
static dim3 nm_grid(const Na2d& p) {
    const int Lx = (p.W + p.d - 1) / p.d, Ly = (p.H + p.d - 1) / p.d;      // the longest residue class
    return dim3(((Lx + NM_TW - 1) / NM_TW) * p.d, ((Ly + NM_TH - 1) / NM_TH) * p.d, p.B * p.nH);
}

int na2d_mfma_fwd(const Na2d& p, int K, hipStream_t stream) {
    const dim3 grid = nm_grid(p);
    UENC_CHECK_ARG(grid.y <= 65535);
    switch (K) {
        case 3: hipLaunchKernelGGL(na2d_mfma_fwd_kernel<3>, grid, dim3(512), 0, stream, p); break;
        case 5: hipLaunchKernelGGL(na2d_mfma_fwd_kernel<5>, grid, dim3(512), 0, stream, p); break;
        case 7: hipLaunchKernelGGL(na2d_mfma_fwd_kernel<7>, grid, dim3(512), 0, stream, p); break;
        default: return UENC_EINVAL;
    }
    UENC_LAUNCH_RET();
}

template <int K>
static int nm_launch_kv(Na2d& p, const dim3& grid, hipStream_t stream) {
    // inverse-neighbourhood extent per axis: T + K + K/2 - 1 beside one border; a class shorter than T + 2K - 1 can touch both
    const int Lx = (p.W + p.d - 1) / p.d, Ly = (p.H + p.d - 1) / p.d;
    p.hh_max = Ly >= NM_TH + 2 * K - 1 ? NM_TH + K + K / 2 - 1 : (Ly < NM_TH + 2 * K - 2 ? Ly : NM_TH + 2 * K - 2);
    p.hw_max = Lx >= NM_TW + 2 * K - 1 ? NM_TW + K + K / 2 - 1 : (Lx < NM_TW + 2 * K - 2 ? Lx : NM_TW + 2 * K - 2);
    const int npos = p.hh_max * p.hw_max + 32;
    const int shm = npos * (64 + 64 + 8) + (2 * K - 1) * (2 * K - 1) * 4;
    static int attr = 0;
    if (attr < shm) {
        const hipError_t e = hipFuncSetAttribute((const void*)na2d_mfma_bwd_kv_kernel<K>, hipFuncAttributeMaxDynamicSharedMemorySize, shm);
        if (e != hipSuccess) return (int)e;
        attr = shm;
    }
    hipLaunchKernelGGL(na2d_mfma_bwd_kv_kernel<K>, grid, dim3(512), shm, stream, p);
    return UENC_OK;
}

int na2d_mfma_bwd(Na2d& p, int K, hipStream_t stream) {
    const dim3 grid = nm_grid(p);
    UENC_CHECK_ARG(grid.y <= 65535);
    int rc;
    switch (K) {
        case 3: hipLaunchKernelGGL(na2d_mfma_bwd_q_kernel<3>, grid, dim3(512), 0, stream, p); rc = nm_launch_kv<3>(p, grid, stream); break;
        case 5: hipLaunchKernelGGL(na2d_mfma_bwd_q_kernel<5>, grid, dim3(512), 0, stream, p); rc = nm_launch_kv<5>(p, grid, stream); break;
        case 7: hipLaunchKernelGGL(na2d_mfma_bwd_q_kernel<7>, grid, dim3(512), 0, stream, p); rc = nm_launch_kv<7>(p, grid, stream); break;
        default: return UENC_EINVAL;
    }
    if (rc != UENC_OK) return rc;
    UENC_LAUNCH_RET();
}
