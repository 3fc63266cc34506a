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
// Schedule.  A workgroup walks `nt` consecutive tiles of one residue class: the halo of tile t + 1 is loaded into registers before the
// arithmetic of tile t and written to LDS after it, so the L2 -> LDS staging (half of every 128-byte line belongs to the neighbouring
// head: the staging runs at the L2's line rate) overlaps the VALU-bound softmax instead of alternating with it, the bias table is
// built once per workgroup, and the grid shrinks from one workgroup per tile (the dispatcher alone took 25 us at 12 288 tiles) to
// a few per CU.  All index arithmetic is 32-bit (the launcher checks H * W * 3C < 2^31; integer multiplies are quarter rate).
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
    static constexpr int NIT = (NPOS + 127) / 128;                       // staging passes: 128 positions x 4 chunks per pass
    static constexpr int TAB = RB * 128;                                 // bias table: [RB][4 shifted copies][32]
};

__device__ __forceinline__ int clampi(int v, int lo, int hi) { return min(max(v, lo), hi); }
__device__ __forceinline__ int fdiv(int a, float inv_b) { return (int)(((float)a + 0.5f) * inv_b); }      // exact for the small counts used here

// workgroup -> (image, head, residue class, tile range); grid (chunks per class, d * d, B * nH)
struct NmWork { int b, h, ry, rx, Ly, Lx, t0, t1; };
__device__ __forceinline__ NmWork nm_work(const Na2d& p) {
    NmWork w;
    w.h = blockIdx.z % p.nH; w.b = blockIdx.z / p.nH;
    w.ry = blockIdx.y / p.d; w.rx = blockIdx.y - w.ry * p.d;
    w.Ly = (p.H - w.ry + p.d - 1) / p.d; w.Lx = (p.W - w.rx + p.d - 1) / p.d;      // this class's extent
    w.t0 = blockIdx.x * p.nt; w.t1 = min(w.t0 + p.nt, p.tiles_y * p.tiles_x);
    return w;
}

// one tile of one residue class (all fields workgroup-uniform)
struct NmTile {
    int ry, rx, py0, px0, Ly, Lx;
    int hsy, hsx, hh, hw;           // halo origin (class positions) and extent
    bool ok;                        // false: the tile lies outside this (ragged) class -- nothing to do
};

// forward / per-query backward: halo = union of the windows of the tile's queries
template <int K>
__device__ __forceinline__ NmTile nm_tile_q(const Na2d& p, const NmWork& w, int t) {
    constexpr int NS = K / 2;
    NmTile g;
    g.ry = w.ry; g.rx = w.rx; g.Ly = w.Ly; g.Lx = w.Lx;
    const int tyi = fdiv(t, p.inv_tiles_x), txi = t - tyi * p.tiles_x;
    g.py0 = tyi * NM_TH; g.px0 = txi * NM_TW;
    g.ok = g.py0 < g.Ly && g.px0 < g.Lx;
    g.hsy = clampi(g.py0 - NS, 0, g.Ly - K); g.hsx = clampi(g.px0 - NS, 0, g.Lx - K);
    g.hh = clampi(min(g.py0 + NM_TH, g.Ly) - 1 - NS, 0, g.Ly - K) + K - g.hsy;
    g.hw = clampi(min(g.px0 + NM_TW, g.Lx) - 1 - NS, 0, g.Lx - K) + K - g.hsx;
    return g;
}

template <int K>
__device__ __forceinline__ int inv_start(int pos) { return pos < K ? 0 : pos - K / 2; }
template <int K>
__device__ __forceinline__ int inv_end(int pos, int L) { return pos >= L - K ? L : pos + K / 2 + 1; }

// per-key backward: halo = union of the inverse neighbourhoods of the tile's keys
template <int K>
__device__ __forceinline__ NmTile nm_tile_kv(const Na2d& p, const NmWork& w, int t) {
    NmTile g;
    g.ry = w.ry; g.rx = w.rx; g.Ly = w.Ly; g.Lx = w.Lx;
    const int tyi = fdiv(t, p.inv_tiles_x), txi = t - tyi * p.tiles_x;
    g.py0 = tyi * NM_TH; g.px0 = txi * NM_TW;
    g.ok = g.py0 < g.Ly && g.px0 < g.Lx;
    const int pyl = min(g.py0 + NM_TH, g.Ly) - 1, pxl = min(g.px0 + NM_TW, g.Lx) - 1;          // last key of the tile
    g.hsy = inv_start<K>(g.py0); g.hsx = inv_start<K>(g.px0);
    g.hh = inv_end<K>(pyl, g.Ly) - g.hsy; g.hw = inv_end<K>(pxl, g.Lx) - g.hsx;
    return g;
}

// Halo staging in two halves: `issue` loads the k and v (or q and dout) chunks of the tile's halo positions into registers, `commit`
// writes them to the LDS images.  Thread -> chunk c = tid & 3 of positions (tid >> 2) + 128 it.  Positions outside the halo are
// zero (they are read as masked operands: must be finite).
template <int NPOS, int HWP, int NIT>
struct NmStage {
    u32x4 a[NIT], b[NIT];
    __device__ __forceinline__ void issue(const NmTile& g, const bf16* src, int d01, unsigned rowstride, unsigned colstride) {
        const int c = threadIdx.x & 3;
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int pos = (threadIdx.x >> 2) + it * 128;
            const int yy = pos / HWP, xx = pos - yy * HWP;
            a[it] = (u32x4){0u, 0u, 0u, 0u}; b[it] = a[it];
            if (g.ok && yy < g.hh && xx < g.hw) {
                const bf16* gp = src + ((unsigned)yy * rowstride + (unsigned)xx * colstride + c * 8);
                a[it] = *(const u32x4*)gp;
                b[it] = *(const u32x4*)(gp + d01);
            }
        }
    }
    __device__ __forceinline__ void commit(unsigned char* img0, unsigned char* img1) const {
        const int c = threadIdx.x & 3;
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int pos = (threadIdx.x >> 2) + it * 128;
            if ((NPOS % 128) == 0 || pos < NPOS) {
                const int off = rm_off(pos, c);
                *(u32x4*)(img0 + off) = a[it];
                *(u32x4*)(img1 + off) = b[it];
            }
        }
    }
};

// bias table [RB rows][4 copies][32]: copy e, position i holds bias column i + e - 8 (zero outside [0, RB)), times log2 e
template <int K>
__device__ __forceinline__ void nm_bias_table(float* tab, const float* rpb, int h) {
    constexpr int RB = 2 * K - 1;
    for (int i = threadIdx.x; i < RB * 128; i += 512) {
        const int row = i >> 7, e = (i >> 5) & 3, col = (i & 31) + e - 8;
        tab[i] = (rpb != nullptr && col >= 0 && col < RB) ? rpb[(h * RB + row) * RB + col] * LOG2E : 0.f;
    }
}

// per-lane description of a wave's 2 x 8 query block inside the workgroup's halo
struct NmQuery {
    bool wave_active, valid;
    unsigned pix;             // this lane's query pixel inside the image b (clamped to a valid one)
    int dy0, dx0;             // this query's window start relative to the wave's key region
    int by0;                  // bias row of key block 0
    int bcol0;                // bias column of key column 0 for the wave's first query column (lane-uniform part of the drpb bins)
    int boff;                 // float offset of this lane's 4 bias columns inside a table row (copy + aligned position)
    unsigned a_rows;          // byte offset of this lane's operand row of key block 0 (frag_rows); block kt: + kt * HWP * 64
    unsigned a_tr[2];         // byte offsets of the transposing reads of key block 0 for output dims 0..15 / 16..31; block kt likewise
};

template <int K>
__device__ __forceinline__ NmQuery nm_query(const Na2d& p, const NmTile& g) {
    constexpr int NS = K / 2, HWP = NmGeom<K>::HWP;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, fr = lane & 15, fg = lane >> 4;
    const int wy = wave >> 1, wx = wave & 1;
    NmQuery q;
    q.wave_active = (g.py0 + 2 * wy < g.Ly) && (g.px0 + 8 * wx < g.Lx);
    const int py_a = min(g.py0 + 2 * wy, g.Ly - 1), px_a = min(g.px0 + 8 * wx, g.Lx - 1);
    const int pyr = g.py0 + 2 * wy + (fr >> 3), pxr = g.px0 + 8 * wx + (fr & 7);
    q.valid = pyr < g.Ly && pxr < g.Lx;
    const int py = min(pyr, g.Ly - 1), px = min(pxr, g.Lx - 1);
    const int sy_a = clampi(py_a - NS, 0, g.Ly - K), sx_a = clampi(px_a - NS, 0, g.Lx - K);
    const int sy = clampi(py - NS, 0, g.Ly - K), sx = clampi(px - NS, 0, g.Lx - K);
    q.dy0 = sy - sy_a; q.dx0 = sx - sx_a;
    const int r0 = (sy_a - g.hsy) * HWP + (sx_a - g.hsx);          // image position of key block 0, key column 0
    q.by0 = sy_a - py + K - 1;
    q.bcol0 = sx_a - px_a + K - 1;
    const int j0 = 4 * fg + sx_a - px + K - 1;                     // bias column of key column 4 fg: in [-7, K + 11]
    q.boff = ((j0 + 8) & 3) * 32 + ((j0 + 8) & ~3);
    // HWP is a multiple of 8 rows, so the row swizzle ((row >> 1) & 3) is the same in every key block: constant block strides
    q.a_rows = rm_off(r0 + fr, fg);
    {
        const int q4 = fr >> 2, p4 = fr & 3;
        const int a0 = rm_off(r0 + 4 * fg + q4, p4 >> 1) + ((p4 & 1) << 3);
        q.a_tr[0] = a0; q.a_tr[1] = a0 ^ 32;
    }
    q.pix = (unsigned)(py * p.d + g.ry) * p.W + px * p.d + g.rx;
    return q;
}

__device__ __forceinline__ f32x4 nm_bias4(const float* tab, int by, int boff) { return *(const f32x4*)(tab + by * 128 + boff); }

template <int HWP>
__device__ __forceinline__ bf16x8 nm_rows(const unsigned char* img, unsigned a_rows, int kt) { return *(const bf16x8*)(img + a_rows + kt * HWP * 64); }

// A operand M^T[dim][k] of the products that sum over two key blocks (kt, kt + 1): k-slots 0..3 <- block kt keys 4 fg + 0..3, 4..7 <- block kt + 1
template <int HWP>
__device__ __forceinline__ bf16x8 nm_tr(const unsigned char* img, unsigned a_tr, int kt) {
    typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4;
    const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(img + a_tr + kt * HWP * 64));
    const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(img + a_tr + (kt + 1) * HWP * 64));
    bf16x8 r;
#pragma unroll
    for (int j = 0; j < 4; ++j) { r[j] = lo[j]; r[4 + j] = hi[j]; }
    return r;
}

__device__ __forceinline__ void nm_store4(bf16* dst, const f32x4& v, float mul) {
    bf16x4 o;
#pragma unroll
    for (int r = 0; r < 4; ++r) o[r] = (bf16)(v[r] * mul);
    *(bf16x4*)dst = o;
}

// ---------------------------------------------------------------------------------------------------------------------
// forward
// ---------------------------------------------------------------------------------------------------------------------
template <int K>
__global__ __launch_bounds__(512) void na2d_mfma_fwd_kernel(Na2d p) {
    typedef NmGeom<K> G;
    constexpr int RB = G::RB, NKT = G::NKT, HWP = G::HWP;
    __shared__ __attribute__((aligned(16))) unsigned char ks[G::NPOS * 64];
    __shared__ __attribute__((aligned(16))) unsigned char vs[G::NPOS * 64];
    __shared__ __attribute__((aligned(16))) float tab[G::TAB];
    const NmWork w = nm_work(p);
    const int C = p.nH * 32;
    const unsigned colstride = p.d * 3 * C, rowstride = colstride * p.W;
    const bf16* qkv_b = p.qkv + (long)w.b * p.H * p.W * 3 * C + w.h * 32;          // this image, this head's q slice
    bf16* out_b = p.out + (long)w.b * p.H * p.W * C + w.h * 32;
    float* lse_b = p.lse ? p.lse + ((long)w.b * p.nH + w.h) * p.H * p.W : nullptr;
    const int lane = threadIdx.x & 63, fg = lane >> 4;
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    const float sc = p.scale * LOG2E;
    nm_bias_table<K>(tab, p.rpb, w.h);
    NmStage<G::NPOS, HWP, G::NIT> st;
    NmTile g = nm_tile_q<K>(p, w, w.t0);
    auto issue = [&](const NmTile& t) {
        st.issue(t, qkv_b + ((unsigned)(t.hsy * p.d + t.ry) * p.W + t.hsx * p.d + t.rx) * (unsigned)(3 * C) + C, C, rowstride, colstride);
    };
    issue(g);
    for (int t = w.t0; t < w.t1; ++t) {
        __syncthreads();                                   // the previous tile's LDS reads are done
        st.commit(ks, vs);
        __syncthreads();
        // this tile's own global operand first: loads return in order, a wait on q would otherwise also wait for the prefetch behind it
        const NmQuery q = nm_query<K>(p, g);
        const bf16x8 qf = *(const bf16x8*)(qkv_b + q.pix * (unsigned)(3 * C) + 8 * fg);
        NmTile gn = g;
        if (t + 1 < w.t1) { gn = nm_tile_q<K>(p, w, t + 1); issue(gn); }      // in flight during the arithmetic below
        if (g.ok && q.wave_active) {                       // whole waves: EXEC stays full for the transposing reads
            bool colok[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) colok[r] = (unsigned)(4 * fg + r - q.dx0) < (unsigned)K;
            f32x4 s[NKT];
            float m = NEG_BIG;
#pragma unroll
            for (int kt = 0; kt < NKT; ++kt) {
                s[kt] = mfma16(nm_rows<HWP>(ks, q.a_rows, kt), qf, zero4);
                const f32x4 bias = nm_bias4(tab, clampi(q.by0 + kt, 0, RB - 1), q.boff);
                const bool rowok = (unsigned)(kt - q.dy0) < (unsigned)K;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    s[kt][r] = (rowok && colok[r]) ? s[kt][r] * sc + bias[r] : NEG_BIG;
                    m = fmaxf(m, s[kt][r]);
                }
            }
            m = xor16_max(m);
            m = xor32_max(m);
            float l = 0.f;
#pragma unroll
            for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    s[kt][r] = fast_exp2(s[kt][r] - m);
                    l += s[kt][r];
                }
            l = xor16_sum(l);
            l = xor32_sum(l);
            f32x4 o[2] = {zero4, zero4};
#pragma unroll
            for (int s2 = 0; s2 < NKT / 2; ++s2) {
                bf16x8 pb;
#pragma unroll
                for (int r = 0; r < 4; ++r) { pb[r] = (bf16)s[2 * s2][r]; pb[4 + r] = (bf16)s[2 * s2 + 1][r]; }
#pragma unroll
                for (int dt = 0; dt < 2; ++dt) o[dt] = mfma16(nm_tr<HWP>(vs, q.a_tr[dt], 2 * s2), pb, o[dt]);
            }
            if (q.valid) {
                const float inv = 1.0f / l;
#pragma unroll
                for (int dt = 0; dt < 2; ++dt) nm_store4(out_b + q.pix * (unsigned)C + dt * 16 + 4 * fg, o[dt], inv);
                if (fg == 0 && lse_b) lse_b[q.pix] = (m + __log2f(l)) * LN2;
            }
        }
        g = gn;
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// backward per query: dq, drpb, delta
// ---------------------------------------------------------------------------------------------------------------------
// drpb.  dS of a key block sits in the lanes as [query (lane & 15)][key column 4 fg + r]; its bias bin is (kt - query row + by_first,
// key column - query column + bcol0) with by_first / bcol0 wave-uniform: a sum along diagonals.  Per key block the wave writes its
// dS tile to LDS as [query][36 floats: 8 zeros, 16 key columns, zeros] (one ds_write_b128 per lane, bank-balanced), and lane L < 46 adds
// the diagonal L % 23 of query row L / 23 (8 unpredicated reads at constant offsets: the zero margins absorb the ends) into a register
// per key block.  Those 8 registers collect over the workgroup's tiles while (by_first, bcol0) stays the same -- every interior tile --
// and go to the workgroup's bins (LDS atomics), then to global memory, once.
#define NM_DS_STRIDE 36
#define NM_DS_WAVE (16 * NM_DS_STRIDE)

template <int K>
__device__ __forceinline__ void nm_drpb_flush(float* dbin, const float (&bacc)[K + 1], int by_first, int bcol0) {
    constexpr int RB = 2 * K - 1, NKT = K + 1, NBIN = 2 * (16 + 7);
    const int lane = threadIdx.x & 63;
    const int bq = lane / 23, bslot = lane - bq * 23;
    const int col = bslot - 7 + bcol0;
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt) {
        const int row = by_first - bq + kt;
        if (lane < NBIN && (unsigned)row < (unsigned)RB && (unsigned)col < (unsigned)RB && bacc[kt] != 0.f) atomicAdd(&dbin[row * RB + col], bacc[kt]);
    }
}

template <int K>
__global__ __launch_bounds__(512) void na2d_mfma_bwd_q_kernel(Na2d p) {
    typedef NmGeom<K> G;
    constexpr int RB = G::RB, NKT = G::NKT, HWP = G::HWP;
    extern __shared__ __attribute__((aligned(16))) unsigned char dynq[];
    unsigned char* ks = dynq;
    unsigned char* vs = ks + G::NPOS * 64;
    float* tab = (float*)(vs + G::NPOS * 64);
    float* dsbuf = tab + G::TAB;                                           // [8 waves][16 queries][36]
    float* dbin = dsbuf + 8 * NM_DS_WAVE;                                  // [RB * RB]
    const NmWork w = nm_work(p);
    const int C = p.nH * 32;
    const unsigned colstride = p.d * 3 * C, rowstride = colstride * p.W;
    const bf16* qkv_b = p.qkv + (long)w.b * p.H * p.W * 3 * C + w.h * 32;
    const bf16* out_b = p.out + (long)w.b * p.H * p.W * C + w.h * 32;
    const bf16* dout_b = p.dout + (long)w.b * p.H * p.W * C + w.h * 32;
    bf16* dq_b = p.dqkv + (long)w.b * p.H * p.W * 3 * C + w.h * 32;
    const float* lse_b = p.lse + ((long)w.b * p.nH + w.h) * p.H * p.W;
    float* delta_b = p.delta + ((long)w.b * p.nH + w.h) * p.H * p.W;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, fr = lane & 15, fg = lane >> 4;
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    const float sc = p.scale * LOG2E;
    nm_bias_table<K>(tab, p.rpb, w.h);
    for (int i = threadIdx.x; i < 8 * NM_DS_WAVE + RB * RB; i += 512) dsbuf[i] = 0.f;       // (dbin follows dsbuf)
    float* buf = dsbuf + wave * NM_DS_WAVE;
    const int bq = lane / 23, bslot = lane - bq * 23;
    const float* bread = buf + (bq & 1) * 8 * NM_DS_STRIDE + bslot + 1;                      // + qx * 37: element (query bq * 8 + qx, column bslot - 7 + qx)
    float* bwrite = buf + fr * NM_DS_STRIDE + 8 + 4 * fg;
    float bacc[NKT];
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt) bacc[kt] = 0.f;
    int acc_by = 0, acc_bc = 0;
    bool acc_any = false;
    NmStage<G::NPOS, HWP, G::NIT> st;
    NmTile g = nm_tile_q<K>(p, w, w.t0);
    auto issue = [&](const NmTile& t) {
        st.issue(t, qkv_b + ((unsigned)(t.hsy * p.d + t.ry) * p.W + t.hsx * p.d + t.rx) * (unsigned)(3 * C) + C, C, rowstride, colstride);
    };
    issue(g);
    for (int t = w.t0; t < w.t1; ++t) {
        __syncthreads();
        st.commit(ks, vs);
        __syncthreads();
        // this tile's own global operands before the prefetch (loads return in order)
        const NmQuery q = nm_query<K>(p, g);
        const bf16x8 qf = *(const bf16x8*)(qkv_b + q.pix * (unsigned)(3 * C) + 8 * fg);
        const bf16x8 dof = *(const bf16x8*)(dout_b + q.pix * (unsigned)C + 8 * fg);
        const bf16x8 ov = *(const bf16x8*)(out_b + q.pix * (unsigned)C + 8 * fg);
        const float lse2 = lse_b[q.pix] * LOG2E;
        NmTile gn = g;
        if (t + 1 < w.t1) { gn = nm_tile_q<K>(p, w, t + 1); issue(gn); }
        if (g.ok && q.wave_active) {
            float delta = 0.f;
#pragma unroll
            for (int c = 0; c < 8; ++c) delta += (float)dof[c] * (float)ov[c];
            delta = xor16_sum(delta);
            delta = xor32_sum(delta);
            if (q.valid && fg == 0) delta_b[q.pix] = delta;
            if (p.drpb) {
                const int by_first = __builtin_amdgcn_readfirstlane(q.by0);       // lane 0 = the wave's first query (row 0, column 0)
                const int bc_first = __builtin_amdgcn_readfirstlane(q.bcol0);
                if (acc_any && (by_first != acc_by || bc_first != acc_bc)) {
                    nm_drpb_flush<K>(dbin, bacc, acc_by, acc_bc);
#pragma unroll
                    for (int kt = 0; kt < NKT; ++kt) bacc[kt] = 0.f;
                }
                acc_by = by_first; acc_bc = bc_first; acc_any = true;
            }
            bool colok[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) colok[r] = q.valid && (unsigned)(4 * fg + r - q.dx0) < (unsigned)K;
            f32x4 o[2] = {zero4, zero4};
#pragma unroll
            for (int s2 = 0; s2 < NKT / 2; ++s2) {
                bf16x8 pb;
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int kt = 2 * s2 + u;
                    const f32x4 sv = mfma16(nm_rows<HWP>(ks, q.a_rows, kt), qf, zero4);
                    const f32x4 dp = mfma16(nm_rows<HWP>(vs, q.a_rows, kt), dof, zero4);
                    const f32x4 bias = nm_bias4(tab, clampi(q.by0 + kt, 0, RB - 1), q.boff);
                    const bool rowok = (unsigned)(kt - q.dy0) < (unsigned)K;
                    f32x4 ds4;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        // (masked through the exponent: a select of the RESULT makes the compiler branch around the exp and its operands)
                        const float e = fast_exp2((rowok && colok[r]) ? sv[r] * sc + bias[r] - lse2 : NEG_BIG);
                        ds4[r] = e * (dp[r] - delta);
                        pb[4 * u + r] = (bf16)ds4[r];
                    }
                    if (p.drpb) {
                        *(f32x4*)bwrite = ds4;
                        __builtin_amdgcn_wave_barrier();
                        float a = 0.f;
#pragma unroll
                        for (int qx = 0; qx < 8; ++qx) a += bread[qx * (NM_DS_STRIDE + 1)];
                        bacc[kt] += a;
                        __builtin_amdgcn_wave_barrier();
                    }
                }
#pragma unroll
                for (int dt = 0; dt < 2; ++dt) o[dt] = mfma16(nm_tr<HWP>(ks, q.a_tr[dt], 2 * s2), pb, o[dt]);
            }
            if (q.valid) {
#pragma unroll
                for (int dt = 0; dt < 2; ++dt) nm_store4(dq_b + q.pix * (unsigned)(3 * C) + dt * 16 + 4 * fg, o[dt], p.scale);
            }
        }
        g = gn;
    }
    if (p.drpb) {
        if (acc_any) nm_drpb_flush<K>(dbin, bacc, acc_by, acc_bc);
        __syncthreads();
        for (int i = threadIdx.x; i < RB * RB; i += 512)
            if (dbin[i] != 0.f) atomicAdd(p.drpb + w.h * RB * RB + i, dbin[i]);
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// backward per key: dk, dv.  Tile of 8 x 16 KEY positions; q, dout, (lse, delta) of the tile's inverse neighbourhood in LDS
// (extents hh_max x hw_max from the launcher: T + K + K/2 - 1 positions per axis beside a border, the whole class when it is short).
// ---------------------------------------------------------------------------------------------------------------------
// bias table of the per-key kernel: [key row - query row + K - 1 + KV_PR][key column - query column + K - 1 + KV_PC], zero margins wide
// enough for every (key, query) pair a wave forms with the padded blocks of its query region, so the index needs no clamp
#define KV_PR 4
#define KV_PC 25
#define KV_ST 48
#define KV_ROWS(K) (2 * (K) - 1 + 2 * KV_PR)

#define NM_KV_PF_NIT 3       // prefetching form: halo positions + 32 <= 3 * 128

template <int K, bool PF>
__global__ __launch_bounds__(512) void na2d_mfma_bwd_kv_kernel(Na2d p) {
    constexpr int RB = 2 * K - 1;
    const int HWQ = p.hw_max, NPOS = p.hh_max * p.hw_max + 32;
    extern __shared__ __attribute__((aligned(16))) unsigned char dyn[];
    unsigned char* qs = dyn;                                  // [NPOS] x 64 B
    unsigned char* gs = qs + NPOS * 64;                       // [NPOS] x 64 B
    float2* ld = (float2*)(gs + NPOS * 64);                   // [NPOS] (lse * log2e, delta)
    float* rpb = (float*)(ld + NPOS);                         // [KV_ROWS][KV_ST]
    const NmWork w = nm_work(p);
    const int C = p.nH * 32;
    const unsigned colq = p.d * 3 * C, rowq = colq * p.W, colg = p.d * C, rowg = colg * p.W, cols = p.d, rows = p.d * p.W;
    const bf16* qkv_b = p.qkv + (long)w.b * p.H * p.W * 3 * C + w.h * 32;
    const bf16* dout_b = p.dout + (long)w.b * p.H * p.W * C + w.h * 32;
    bf16* dqkv_b = p.dqkv + (long)w.b * p.H * p.W * 3 * C + w.h * 32;
    const float* lse_b = p.lse + ((long)w.b * p.nH + w.h) * p.H * p.W;
    const float* delta_b = p.delta + ((long)w.b * p.nH + w.h) * p.H * p.W;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, fr = lane & 15, fg = lane >> 4;
    const int wy = wave >> 1, wx = wave & 1;
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    const float sc = p.scale * LOG2E;
    for (int i = threadIdx.x; i < KV_ROWS(K) * KV_ST; i += 512) {
        const int row = i / KV_ST - KV_PR, col = i % KV_ST - KV_PC;
        rpb[i] = (p.rpb != nullptr && (unsigned)row < (unsigned)RB && (unsigned)col < (unsigned)RB) ? p.rpb[(w.h * RB + row) * RB + col] * LOG2E : 0.f;
    }
    const float inv_hwq = 1.0f / (float)HWQ;
    const int c = threadIdx.x & 3;
    // staging: q and dout chunks and (chunk-0 lanes) the statistics of halo position (tid >> 2) + 128 it; zero outside the halo.
    // PF: the loads of tile t + 1 are issued before the arithmetic of tile t and committed to LDS after it (3 passes in registers).
    constexpr int NIT = PF ? NM_KV_PF_NIT : 1;
    u32x4 sq[NIT], sg[NIT];
    float2 sst[NIT];
    auto load_pos = [&](const NmTile& t, unsigned pix0, int pos, u32x4& vq, u32x4& vg, float2& st) {
        const int yy = fdiv(pos, inv_hwq), xx = pos - yy * HWQ;
        vq = (u32x4){0u, 0u, 0u, 0u}; vg = vq; st = (float2){0.f, 0.f};
        if (t.ok && pos < NPOS && yy < t.hh && xx < t.hw) {
            vq = *(const u32x4*)(qkv_b + (pix0 * (unsigned)(3 * C) + (unsigned)yy * rowq + (unsigned)xx * colq + c * 8));
            vg = *(const u32x4*)(dout_b + (pix0 * (unsigned)C + (unsigned)yy * rowg + (unsigned)xx * colg + c * 8));
            if (c == 0) {
                const unsigned so = pix0 + (unsigned)yy * rows + (unsigned)xx * cols;
                st.x = lse_b[so] * LOG2E; st.y = delta_b[so];
            }
        }
    };
    auto store_pos = [&](int pos, const u32x4& vq, const u32x4& vg, const float2& st) {
        if (pos < NPOS) {
            const int off = rm_off(pos, c);
            *(u32x4*)(qs + off) = vq;
            *(u32x4*)(gs + off) = vg;
            if (c == 0) ld[pos] = st;
        }
    };
    auto issue = [&](const NmTile& t) {
        const unsigned pix0 = (unsigned)(t.hsy * p.d + t.ry) * p.W + t.hsx * p.d + t.rx;
#pragma unroll
        for (int it = 0; it < NIT; ++it) load_pos(t, pix0, (threadIdx.x >> 2) + it * 128, sq[it], sg[it], sst[it]);
    };
    NmTile g = nm_tile_kv<K>(p, w, w.t0);
    if (PF) issue(g);
    for (int t = w.t0; t < w.t1; ++t) {
        __syncthreads();                                       // the previous tile's LDS reads are done
        if (PF) {
#pragma unroll
            for (int it = 0; it < NIT; ++it) store_pos((threadIdx.x >> 2) + it * 128, sq[it], sg[it], sst[it]);
        } else {
            const unsigned pix0 = (unsigned)(g.hsy * p.d + g.ry) * p.W + g.hsx * p.d + g.rx;
            for (int pos = threadIdx.x >> 2; pos < NPOS; pos += 128) {
                load_pos(g, pix0, pos, sq[0], sg[0], sst[0]);
                store_pos(pos, sq[0], sg[0], sst[0]);
            }
        }
        __syncthreads();
        // this tile's own global operands before the prefetch (loads return in order)
        const int kyr = g.py0 + 2 * wy + (fr >> 3), kxr = g.px0 + 8 * wx + (fr & 7);
        const bool kvalid = kyr < g.Ly && kxr < g.Lx;
        const int ky = clampi(kyr, 0, g.Ly - 1), kx = clampi(kxr, 0, g.Lx - 1);
        const unsigned pix = (unsigned)(ky * p.d + g.ry) * p.W + kx * p.d + g.rx;
        const bf16x8 kf = *(const bf16x8*)(qkv_b + pix * (unsigned)(3 * C) + C + 8 * fg);
        const bf16x8 vf = *(const bf16x8*)(qkv_b + pix * (unsigned)(3 * C) + 2 * C + 8 * fg);
        NmTile gn = g;
        if (t + 1 < w.t1) { gn = nm_tile_kv<K>(p, w, t + 1); if (PF) issue(gn); }
        if (g.ok && g.py0 + 2 * wy < g.Ly && g.px0 + 8 * wx < g.Lx) {          // whole waves
            // a query attends this lane's key iff it lies in the key's inverse neighbourhood (exact, per axis)
            const int qys_k = inv_start<K>(ky), qyn_k = inv_end<K>(ky, g.Ly) - qys_k;
            const int qxs_k = inv_start<K>(kx), qxn_k = inv_end<K>(kx, g.Lx) - qxs_k;
            // the wave's query region (wave-uniform): union of the inverse neighbourhoods of its keys
            const int ky_a = min(g.py0 + 2 * wy, g.Ly - 1), ky_b = min(g.py0 + 2 * wy + 1, g.Ly - 1);
            const int kx_a = min(g.px0 + 8 * wx, g.Lx - 1), kx_b = min(g.px0 + 8 * wx + 7, g.Lx - 1);
            const int qy0 = inv_start<K>(ky_a), qy1 = inv_end<K>(ky_b, g.Ly);
            const int qx0 = inv_start<K>(kx_a), qx1 = inv_end<K>(kx_b, g.Lx);
            const int ncb = (qx1 - qx0 + 15) >> 4, nblk = (qy1 - qy0) * ncb;
            f32x4 dk[2] = {zero4, zero4}, dv[2] = {zero4, zero4};
            for (int j = 0; j < nblk; j += 2) {
                int r0[2];
                f32x4 pr[2], ds[2];
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const bool real = j + u < nblk;
                    const int blk = real ? j + u : 0;
                    const int row = ncb == 2 ? blk >> 1 : blk, cb = blk - row * ncb;
                    const int qy = qy0 + row, bx = qx0 + 16 * cb;
                    r0[u] = (qy - g.hsy) * HWQ + (bx - g.hsx);
                    const f32x4 sv = mfma16(frag_rows(qs, r0[u], fr, fg), kf, zero4);
                    const f32x4 dp = mfma16(frag_rows(gs, r0[u], fr, fg), vf, zero4);
                    const bool rowok = real && (unsigned)(qy - qys_k) < (unsigned)qyn_k;
                    const int cbase = bx + 4 * fg - qxs_k;                                      // query column 4 fg + r relative to the key's first query
                    const float* brow = rpb + (ky - qy + K - 1 + KV_PR) * KV_ST + (kx - bx - 4 * fg + K - 1 + KV_PC) - 3;
                    const float2* lrow = ld + r0[u] + 4 * fg;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const bool ok = rowok && (unsigned)(cbase + r) < (unsigned)qxn_k;
                        const float2 stq = lrow[r];
                        const float arg = sv[r] * sc + brow[3 - r] - stq.x;                  // loads outside the select: no branch around them
                        pr[u][r] = fast_exp2(ok ? arg : NEG_BIG);
                        ds[u][r] = pr[u][r] * (dp[r] - stq.y);
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
            if (kvalid) {
#pragma unroll
                for (int dt = 0; dt < 2; ++dt) {
                    nm_store4(dqkv_b + pix * (unsigned)(3 * C) + C + dt * 16 + 4 * fg, dk[dt], p.scale);
                    nm_store4(dqkv_b + pix * (unsigned)(3 * C) + 2 * C + dt * 16 + 4 * fg, dv[dt], 1.0f);
                }
            }
        }
        g = gn;
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------------------------------------------------
static dim3 nm_plan(Na2d& p) {
    const int Lx = (p.W + p.d - 1) / p.d, Ly = (p.H + p.d - 1) / p.d;      // the longest residue class
    p.tiles_x = (Lx + NM_TW - 1) / NM_TW; p.tiles_y = (Ly + NM_TH - 1) / NM_TH;
    p.inv_tiles_x = 1.0f / (float)p.tiles_x;
    const int per_class = p.tiles_x * p.tiles_y;
    const long total = (long)per_class * p.d * p.d * p.B * p.nH;
    // tiles per workgroup: enough workgroups for ~6 per CU, at most 8 tiles each, never more than a class has
    int nt = (int)(total / 1536);
    nt = nt < 1 ? 1 : (nt > 8 ? 8 : nt);
    if (nt > per_class) nt = per_class;
    const char* ev = getenv("UENC_NA2D_NT");
    if (ev && atoi(ev) > 0) nt = atoi(ev);
    p.nt = nt;
    return dim3((per_class + nt - 1) / nt, p.d * p.d, p.B * p.nH);
}

int na2d_mfma_fwd(const Na2d& p0, int K, hipStream_t stream) {
    Na2d p = p0;
    const dim3 grid = nm_plan(p);
    UENC_CHECK_ARG(grid.y <= 65535 && grid.z <= 65535);
    switch (K) {
        case 3: hipLaunchKernelGGL(na2d_mfma_fwd_kernel<3>, grid, dim3(512), 0, stream, p); break;
        case 5: hipLaunchKernelGGL(na2d_mfma_fwd_kernel<5>, grid, dim3(512), 0, stream, p); break;
        case 7: hipLaunchKernelGGL(na2d_mfma_fwd_kernel<7>, grid, dim3(512), 0, stream, p); break;
        default: return UENC_EINVAL;
    }
    UENC_LAUNCH_RET();
}

// LDS extent of the per-key kernel: the largest inverse-neighbourhood halo over all tiles of all residue classes, per axis
// (T + K - 1 positions for an interior tile, up to T + K + K/2 - 1 where a tile's last key is the first of the clamped border band)
static int nm_axis_extent(int len, int d, int K, int T) {
    int best = 0;
    for (int r = 0; r < d && r < len; ++r) {
        const int L = (len - r + d - 1) / d;
        for (int p0 = 0; p0 < L; p0 += T) {
            const int pl = (p0 + T < L ? p0 + T : L) - 1;
            const int s = p0 < K ? 0 : p0 - K / 2, e = pl >= L - K ? L : pl + K / 2 + 1;
            if (e - s > best) best = e - s;
        }
    }
    return best;
}
static void nm_kv_extents(Na2d& p, int K) {
    p.hh_max = nm_axis_extent(p.H, p.d, K, NM_TH);
    p.hw_max = nm_axis_extent(p.W, p.d, K, NM_TW);
}

bool na2d_mfma_supported(int H, int W, int nH, int K, int dilation) {
    if (K > 7 || (long)H * W * 3 * nH * 32 >= (1L << 31)) return false;
    Na2d p = {};
    p.H = H; p.W = W; p.d = dilation;
    nm_kv_extents(p, K);
    return true;
}

template <int K, bool PF>
static int nm_launch_kv2(Na2d& p, const dim3& grid, int shm, hipStream_t stream) {
    static int attr = 0;
    if (attr < shm) {
        const hipError_t e = hipFuncSetAttribute((const void*)na2d_mfma_bwd_kv_kernel<K, PF>, hipFuncAttributeMaxDynamicSharedMemorySize, shm);
        if (e != hipSuccess) return (int)e;
        attr = shm;
    }
    hipLaunchKernelGGL((na2d_mfma_bwd_kv_kernel<K, PF>), grid, dim3(512), shm, stream, p);
    return UENC_OK;
}

template <int K>
static int nm_launch_kv(Na2d& p, const dim3& grid, hipStream_t stream) {
    const int npos = p.hh_max * p.hw_max + 32;
    const int shm = npos * (64 + 64 + 8) + KV_ROWS(K) * KV_ST * 4;
    const char* ev = getenv("UENC_NA2D_VARIANT");
    const bool pf = npos <= NM_KV_PF_NIT * 128 && p.nt > 1 && ev && (atoi(ev) & 4);      // A/B only: the prefetching form needs 153 VGPRs (one workgroup per CU) and measured slower
    return pf ? nm_launch_kv2<K, true>(p, grid, shm, stream) : nm_launch_kv2<K, false>(p, grid, shm, stream);
}

template <int K>
static int nm_launch_q(Na2d& p, const dim3& grid, hipStream_t stream) {
    typedef NmGeom<K> G;
    const int shm = 2 * G::NPOS * 64 + (G::TAB + 8 * NM_DS_WAVE + G::RB * G::RB) * 4;
    static bool attr = false;
    if (!attr) {
        const hipError_t e = hipFuncSetAttribute((const void*)na2d_mfma_bwd_q_kernel<K>, hipFuncAttributeMaxDynamicSharedMemorySize, shm);
        if (e != hipSuccess) return (int)e;
        attr = true;
    }
    hipLaunchKernelGGL(na2d_mfma_bwd_q_kernel<K>, grid, dim3(512), shm, stream, p);
    return UENC_OK;
}

int na2d_mfma_bwd(Na2d& p, int K, hipStream_t stream) {
    const dim3 grid = nm_plan(p);
    UENC_CHECK_ARG(grid.y <= 65535 && grid.z <= 65535);
    nm_kv_extents(p, K);
    int rc;
    switch (K) {
        case 3: rc = nm_launch_q<3>(p, grid, stream); if (rc == UENC_OK) rc = nm_launch_kv<3>(p, grid, stream); break;
        case 5: rc = nm_launch_q<5>(p, grid, stream); if (rc == UENC_OK) rc = nm_launch_kv<5>(p, grid, stream); break;
        case 7: rc = nm_launch_q<7>(p, grid, stream); if (rc == UENC_OK) rc = nm_launch_kv<7>(p, grid, stream); break;
        default: return UENC_EINVAL;
    }
    if (rc != UENC_OK) return rc;
    UENC_LAUNCH_RET();
}
