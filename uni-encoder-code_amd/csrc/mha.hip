// Multi-head attention cores of the OneFormer transformer decoder (head_dim 32), forward and backward (gfx950).
//
// Replaces the core of every nn.MultiheadAttention of the decoder — scaled q k^T, optional boolean
// mask shared by all heads (True = blocked), softmax, @ v:
//   self-attention over the 150 queries            transformer_decoder/oneformer_transformer_decoder.py:63-67
//   masked cross-attention, 2k / 8k / 32k keys     :122-127 with the mask of :504-511
//   class-transformer cross-attention, 131 072 keys transformer_decoder/transformer.py:268-297 via :434-438
// (the projections around it are GEMMs).  Few queries (<= 160 per launch slice), many keys: the key
// range is split over workgroups (flash-decoding style) so that all CUs stream K / V, with an
// online softmax per (query, split) and a combine kernel.
//
// Layout: q (B, Lq, *) / k, v (B, S, *) bf16, heads interleaved in the row (head h at columns 32h..32h+31),
// arbitrary row / batch strides (slices of packed projections are consumed in place).
// One workgroup = 4 waves = one (batch, head, key range).  Q (and dO) sit in LDS for the whole
// workgroup; K / V stream through double-buffered 64-key LDS blocks (16-byte coalesced loads, XOR-
// swizzled 64-byte rows as in window_attn.hip).  Forward / dQ: a wave owns query tiles (S^T = K Q^T
// on MFMA: a lane owns one query column, softmax state in registers, P^T feeds the PV MFMA from
// registers, V^T / K^T via the transposing LDS read).  dK / dV: a wave owns a 16-key tile and sweeps
// the queries (S = Q K^T orientation), so neither needs a cross-workgroup reduction; dQ partials of the
// key splits are added with fp32 atomics (160 x 32 floats per split).
// Algorithmic HBM bytes: (Lq + 2 S) * 64 per (batch, head) forward; mask Lq * S bytes when present.
#include "common.h"
#include "lds_frag.h"

#define LOG2E 1.4426950408889634f
// v_max without the canonicalising v_max x, x that fmaxf() carries for signalling NaNs (the scores are finite or -inf here)
__device__ __forceinline__ float vmaxf(float a, float b) { float r; asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ float vmax3f(float a, float b, float c) { float r; asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }
#define NEG_INF (-__builtin_huge_valf())
#define MQ 160            // queries per launch slice (10 MFMA tiles)
#define MQT 10
#define KB 64             // keys per LDS block

struct MhaP {
    const bf16 *q, *k, *v;
    long q_bs, q_rs, k_bs, k_rs, v_bs, v_rs;       // batch / row strides in elements
    const unsigned char* mask; long mask_rs;        // (B, Lq, mask_rs >= S) or null; nonzero = blocked
    bf16* out; long o_bs, o_rs;                     // (B, Lq, *)
    float* lse;                                     // (B, H, Lq) log2-domain log-sum-exp
    float *ws_m, *ws_l, *ws_o;                      // split partials: [(b*H+h)*nsplit + s][MQ], [..][MQ], [..][MQ][32]
    // backward
    const bf16* dout; long do_bs, do_rs;
    float* dq;  long dq_bs, dq_rs;                  // fp32, accumulated (caller zeroes)
    bf16 *dk, *dv; long dk_bs, dk_rs, dv_bs, dv_rs;
    int B, H, Lq, S, nsplit, keys_per_split;
    float scale;
    unsigned drop_thresh, seed;                     // attention-probability dropout (0 = off): attn_keep() of common.h
    float inv_keep;
};

__device__ __forceinline__ void stage_rows(unsigned char* img, const bf16* base, long rs, int row0, int nrows, int nvalid,
                                           int nthreads) {
    // rows [row0, row0 + nrows) of a (.., 32)-wide head slice -> LDS image rows 0..nrows-1; rows >= nvalid are zero
    for (int idx = threadIdx.x; idx < nrows * 4; idx += nthreads) {
        const int r = idx >> 2, ch = idx & 3;
        u32x4 v = {0u, 0u, 0u, 0u};
        if (row0 + r < nvalid) v = *(const u32x4*)(base + (long)(row0 + r) * rs + ch * 8);
        *(u32x4*)(img + rm_off(r, ch)) = v;
    }
}

// Block -> (key split, image * H + head).  A head reads 64 bytes of every 512-byte K / V row: heads 2i and 2i + 1 split each
// 128-byte line.  Blocks are numbered so that the two heads of a pair, for the same image and key range, are CONSECUTIVE blocks
// of one XCD (hardware block b runs on XCD slot b % 8): they stream the same lines at the same time and the second read of every
// line is an L2 hit instead of a second trip to the fabric.  (Odd head counts keep the plain order.)  Speed only.
__device__ __forceinline__ bool mha_decode_block(const MhaP& p, int& split, int& bh) {
    const int total = p.nsplit * p.B * p.H;
    if (p.H & 1) {
        if ((int)blockIdx.x >= total) return false;
        split = blockIdx.x % p.nsplit; bh = blockIdx.x / p.nsplit;
        return true;
    }
    const int x = blockIdx.x & 7, j = blockIdx.x >> 3, npair = p.H >> 1;
    const int u = (j >> 1) * 8 + x;                       // unit = (image, key split, head pair), head pair fastest
    if (u >= p.nsplit * p.B * npair) return false;
    const int hp = u % npair, rest = u / npair;
    split = rest % p.nsplit;
    bh = (rest / p.nsplit) * p.H + 2 * hp + (j & 1);
    return true;
}

// ------------------------------------------------------------------------------------------------
// forward (MODE 0) and dQ (MODE 1): waves own query tiles, loop over the key range
// ------------------------------------------------------------------------------------------------
template <int MODE, bool DROP>
__global__ __launch_bounds__(256, 2) void mha_q_kernel(MhaP p) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[(MQ + MQ + 4 * KB) * 64];
    unsigned char* Qs = smem;
    unsigned char* dOs = smem + MQ * 64;                 // dQ mode only
    unsigned char* Ks = smem + 2 * MQ * 64;              // 2 stages x 64 rows
    unsigned char* Vs = Ks + 2 * KB * 64;

    int split, bh;
    if (!mha_decode_block(p, split, bh)) return;
    const int q0 = blockIdx.z * MQ;
    const int b = bh / p.H, h = bh - b * p.H;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, fr = lane & 15, fg = lane >> 4;
    const int k_begin = split * p.keys_per_split;
    const int k_end = min(p.S, k_begin + p.keys_per_split);
    const int nblk = (k_end - k_begin + KB - 1) / KB;
    const bf16* qb = p.q + b * p.q_bs + h * 32;
    const bf16* kb = p.k + b * p.k_bs + h * 32;
    const bf16* vb = p.v + b * p.v_bs + h * 32;

    stage_rows(Qs, qb, p.q_rs, q0, MQ, p.Lq, 256);
    if (MODE == 1) stage_rows(dOs, p.dout + b * p.do_bs + h * 32, p.do_rs, q0, MQ, p.Lq, 256);
    // K / V block staging: one 16-byte chunk of each per thread
    const int srow = threadIdx.x >> 2, sch = threadIdx.x & 3;
    u32x4 rk, rv;
    auto gload = [&](int blk) {
        const int key = k_begin + blk * KB + srow;
        rk = (u32x4){0u, 0u, 0u, 0u}; rv = rk;
        if (key < k_end) {
            rk = *(const u32x4*)(kb + (long)key * p.k_rs + sch * 8);
            rv = *(const u32x4*)(vb + (long)key * p.v_rs + sch * 8);
        }
    };
    auto lstore = [&](int st) {
        *(u32x4*)(Ks + st * KB * 64 + rm_off(srow, sch)) = rk;
        *(u32x4*)(Vs + st * KB * 64 + rm_off(srow, sch)) = rv;
    };
    if (nblk > 0) { gload(0); lstore(0); }
    __syncthreads();

    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    const float sc = p.scale * LOG2E;
    // per owned query tile (qt = wave + 4*i): state
    bf16x8 qf[3], dof[3];
    float m_run[3], l_run[3], lse_q[3], dl_q[3];
    f32x4 o[3][2];
    int qidx[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int qt = wave + 4 * i;
        qidx[i] = q0 + qt * 16 + fr;
        m_run[i] = -1e30f; l_run[i] = 0.f; o[i][0] = zero4; o[i][1] = zero4; lse_q[i] = 0.f; dl_q[i] = 0.f;
        if (qt < MQT) {
            qf[i] = frag_rows(Qs, qt * 16, fr, fg);
            if (MODE == 1) {
                dof[i] = frag_rows(dOs, qt * 16, fr, fg);
                float part = 0.f;
                if (qidx[i] < p.Lq) {
                    const bf16x8 ov = *(const bf16x8*)(p.out + b * p.o_bs + (long)qidx[i] * p.o_rs + h * 32 + 8 * fg);
#pragma unroll
                    for (int j = 0; j < 8; ++j) part += (float)ov[j] * (float)dof[i][j];
                    lse_q[i] = p.lse[((long)b * p.H + h) * p.Lq + qidx[i]];
                }
                part = xor16_sum(part);
                part = xor32_sum(part);
                dl_q[i] = part;
            }
        }
    }

    for (int blk = 0; blk < nblk; ++blk) {
        const int st = blk & 1;
        if (blk + 1 < nblk) gload(blk + 1);
        const unsigned char* Kc = Ks + st * KB * 64;
        const unsigned char* Vc = Vs + st * KB * 64;
        const int kbase = k_begin + blk * KB;
        // forward: the mask words of all of this wave's query tiles are requested at the top of the block (one L2 round trip per block, under
        // the first tile's MFMAs, instead of one per query tile: -3 %); the dQ form, with twice the live state, measured 20 % SLOWER that
        // way and keeps the per-tile loads
        unsigned mka[3][4];
#pragma unroll
        for (int i = 0; i < 3; ++i) {
#pragma unroll
            for (int kt = 0; kt < 4; ++kt) mka[i][kt] = 0u;
            if (MODE == 0 && p.mask != nullptr && wave + 4 * i < MQT && qidx[i] < p.Lq) {
                const unsigned char* mrow = p.mask + ((long)b * p.Lq + qidx[i]) * p.mask_rs + kbase + 4 * fg;
#pragma unroll
                for (int kt = 0; kt < 4; ++kt)
                    if (kbase + kt * 16 + 4 * fg < k_end) mka[i][kt] = *(const unsigned*)(mrow + kt * 16);   // row padded to a multiple of 4
            }
        }
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int qt = wave + 4 * i;
            if (qt >= MQT) continue;
            f32x4 s[4];
#pragma unroll
            for (int kt = 0; kt < 4; ++kt) s[kt] = mfma16(frag_rows(Kc, kt * 16, fr, fg), qf[i], zero4);
            // validity: key < k_end and not blocked by the mask
            unsigned mk[4] = {mka[i][0], mka[i][1], mka[i][2], mka[i][3]};
            if (MODE == 1 && p.mask != nullptr && qidx[i] < p.Lq) {
                const unsigned char* mrow = p.mask + ((long)b * p.Lq + qidx[i]) * p.mask_rs + kbase + 4 * fg;
#pragma unroll
                for (int kt = 0; kt < 4; ++kt)
                    if (kbase + kt * 16 + 4 * fg < k_end) mk[kt] = *(const unsigned*)(mrow + kt * 16);
            }
            // The softmax arithmetic was 12.5 VALU instructions per score (260 per query tile and block: scale, two compares and an AND for
            // validity, select + canonicalising max, subtract, exp, select, add ...) and the kernels are issue-bound on it.  Now a blocked or
            // out-of-range score is set to -inf ONCE (and not at all in a full, unmasked block), the running maximum is taken on the raw
            // scores (the scale is positive), and exp2(fma(s, scale, -max)) needs no select: exp2(-inf) = 0.
            const bool hasmask = p.mask != nullptr, fullblk = kbase + KB <= k_end;         // wave-uniform
            float v[4][4];                              // (a copy in plain registers: the MFMA results live in accumulation registers)
#pragma unroll
            for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                for (int r = 0; r < 4; ++r) v[kt][r] = s[kt][r];
            if (hasmask || !fullblk) {
#pragma unroll
                for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int key = kbase + kt * 16 + 4 * fg + r;
                        const bool okk = (key < k_end) && (((mk[kt] >> (8 * r)) & 0xffu) == 0u);
                        v[kt][r] = okk ? v[kt][r] : NEG_INF;
                    }
            }
            if (MODE == 0) {
                float mloc = vmax3f(v[0][0], v[0][1], v[0][2]);
                mloc = vmax3f(mloc, v[0][3], v[1][0]); mloc = vmax3f(mloc, v[1][1], v[1][2]); mloc = vmax3f(mloc, v[1][3], v[2][0]);
                mloc = vmax3f(mloc, v[2][1], v[2][2]); mloc = vmax3f(mloc, v[2][3], v[3][0]); mloc = vmax3f(mloc, v[3][1], v[3][2]);
                mloc = vmaxf(mloc, v[3][3]);
                mloc = xor16_max(mloc);
                mloc = xor32_max(mloc);
                const float mnew = vmaxf(m_run[i], mloc * sc);        // (-inf * sc = -inf: a block with nothing visible keeps the running maximum)
                const float alpha = fast_exp2(m_run[i] - mnew);
                const float nm = -mnew;
                float sum = 0.f;
#pragma unroll
                for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float e = fast_exp2(__builtin_fmaf(v[kt][r], sc, nm));
                        sum += e;                                    // the softmax normaliser is taken BEFORE dropout
                        v[kt][r] = e;
                        if (DROP)
                            v[kt][r] = attn_keep(p.seed, p.drop_thresh, ((unsigned long long)bh * p.Lq + qidx[i]) * p.S + (kbase + kt * 16 + 4 * fg + r))
                                           ? e * p.inv_keep : 0.f;
                    }
                sum = xor16_sum(sum);
                sum = xor32_sum(sum);
                l_run[i] = l_run[i] * alpha + sum;
                m_run[i] = mnew;
#pragma unroll
                for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) o[i][dt][r] *= alpha;
#pragma unroll
                for (int kb2 = 0; kb2 < 2; ++kb2) {
                    bf16x8 pb;
#pragma unroll
                    for (int r = 0; r < 4; ++r) { pb[r] = (bf16)v[2 * kb2][r]; pb[4 + r] = (bf16)v[2 * kb2 + 1][r]; }
#pragma unroll
                    for (int dt = 0; dt < 2; ++dt)
                        o[i][dt] = mfma16(frag_tr(Vc, 32 * kb2, 32 * kb2 + 16, dt * 16, lane), pb, o[i][dt]);
                }
            } else {
                // dS^T = P * (dP^T - delta), P from the saved log-sum-exp; dQ^T += K^T dS^T
#pragma unroll
                for (int kt = 0; kt < 4; ++kt) {
                    const f32x4 dp = mfma16(frag_rows(Vc, kt * 16, fr, fg), dof[i], zero4);
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float pr = fast_exp2(__builtin_fmaf(v[kt][r], sc, -lse_q[i]));
                        float dpv = dp[r];
                        if (DROP)
                            dpv = attn_keep(p.seed, p.drop_thresh, ((unsigned long long)bh * p.Lq + qidx[i]) * p.S + (kbase + kt * 16 + 4 * fg + r))
                                      ? dpv * p.inv_keep : 0.f;
                        v[kt][r] = pr * (dpv - dl_q[i]);
                    }
                }
#pragma unroll
                for (int kb2 = 0; kb2 < 2; ++kb2) {
                    bf16x8 pb;
#pragma unroll
                    for (int r = 0; r < 4; ++r) { pb[r] = (bf16)v[2 * kb2][r]; pb[4 + r] = (bf16)v[2 * kb2 + 1][r]; }
#pragma unroll
                    for (int dt = 0; dt < 2; ++dt)
                        o[i][dt] = mfma16(frag_tr(Kc, 32 * kb2, 32 * kb2 + 16, dt * 16, lane), pb, o[i][dt]);
                }
            }
        }
        if (blk + 1 < nblk) lstore(st ^ 1);
        __syncthreads();
    }

    // write-out
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int qt = wave + 4 * i;
        if (qt >= MQT || qidx[i] >= p.Lq) continue;
        if (MODE == 1) {
            float* dst = p.dq + b * p.dq_bs + (long)qidx[i] * p.dq_rs + h * 32;
#pragma unroll
            for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                for (int r = 0; r < 4; ++r) atomicAdd(dst + dt * 16 + 4 * fg + r, o[i][dt][r] * p.scale);
        } else if (p.nsplit == 1) {
            const float inv = l_run[i] > 0.f ? 1.0f / l_run[i] : 0.f;
#pragma unroll
            for (int dt = 0; dt < 2; ++dt) {
                bf16x4 ov;
#pragma unroll
                for (int r = 0; r < 4; ++r) ov[r] = (bf16)(o[i][dt][r] * inv);
                *(bf16x4*)(p.out + b * p.o_bs + (long)qidx[i] * p.o_rs + h * 32 + dt * 16 + 4 * fg) = ov;
            }
            if (fg == 0 && p.lse != nullptr) p.lse[((long)b * p.H + h) * p.Lq + qidx[i]] = m_run[i] + log2f(fmaxf(l_run[i], 1e-37f));
        } else {
            const long part = ((long)bh * gridDim.z + blockIdx.z) * p.nsplit + split;
            const int ql = qt * 16 + fr;
            if (fg == 0) { p.ws_m[part * MQ + ql] = m_run[i]; p.ws_l[part * MQ + ql] = l_run[i]; }
#pragma unroll
            for (int dt = 0; dt < 2; ++dt)
                *(f32x4*)(p.ws_o + (part * MQ + ql) * 32 + dt * 16 + 4 * fg) = o[i][dt];
        }
    }
}

// combine the key-range partials: one thread per (slice, query, d)
__global__ __launch_bounds__(256) void mha_combine_kernel(MhaP p, int nslices) {
    const long gid = (long)blockIdx.x * 256 + threadIdx.x;
    const int d = (int)(gid & 31);
    const long qg = gid >> 5;                     // (bh * nslices + slice) * MQ + ql
    const int ql = (int)(qg % MQ);
    const long bs = qg / MQ;
    const int slice = (int)(bs % nslices);
    const long bh = bs / nslices;
    if (bh >= (long)p.B * p.H) return;
    const int qi = slice * MQ + ql;
    if (qi >= p.Lq) return;
    const int b = (int)(bh / p.H), h = (int)(bh - (long)b * p.H);
    const long base = bs * p.nsplit;
    float M = -1e30f;
    for (int s = 0; s < p.nsplit; ++s) M = fmaxf(M, p.ws_m[(base + s) * MQ + ql]);
    float L = 0.f, O = 0.f;
    for (int s = 0; s < p.nsplit; ++s) {
        const float w = fast_exp2(p.ws_m[(base + s) * MQ + ql] - M);
        L += p.ws_l[(base + s) * MQ + ql] * w;
        O += p.ws_o[((base + s) * MQ + ql) * 32 + d] * w;
    }
    p.out[b * p.o_bs + (long)qi * p.o_rs + h * 32 + d] = (bf16)(L > 0.f ? O / L : 0.f);
    if (d == 0 && p.lse != nullptr) p.lse[((long)b * p.H + h) * p.Lq + qi] = M + log2f(fmaxf(L, 1e-37f));
}

// ------------------------------------------------------------------------------------------------
// dK / dV: a wave owns a 16-key tile of each 64-key block and sweeps all queries
// ------------------------------------------------------------------------------------------------
template <bool DROP>
__global__ __launch_bounds__(256, 2) void mha_dkdv_kernel(MhaP p) {
    constexpr int NQB = MQT / 2;     // 32-query blocks
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int nslices = (p.Lq + MQ - 1) / MQ;
    unsigned char* Qs = smem;                                   // nslices x MQ rows
    unsigned char* dOs = Qs + (long)nslices * MQ * 64;
    unsigned char* Ks = dOs + (long)nslices * MQ * 64;         // 2 stages
    unsigned char* Vs = Ks + 2 * KB * 64;
    float* lse = (float*)(Vs + 2 * KB * 64);
    float* delta = lse + nslices * MQ;
    // Mask tile of the current 64-key block, TRANSPOSED: row = key, NQ + 4 bytes per row (an odd number of dwords: the 16 keys of a wave
    // fall into 16 banks), byte q = 1 when query q may not see the key (rows of the padding queries are all 1).  A lane then reads the
    // mask of its key for four consecutive queries as ONE LDS dword; loading the bytes from memory one by one inside the score loop
    // (8 dependent L2 round trips per 32 queries, each behind its own s_waitcnt) made the masked launches 2.4 x slower per key than
    // the unmasked ones.
    unsigned char* Ms = (unsigned char*)(delta + nslices * MQ);
    const int MSTR = nslices * MQ + 4;

    int split, bh;
    if (!mha_decode_block(p, split, bh)) return;
    const int b = bh / p.H, h = bh - b * p.H;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, fr = lane & 15, fg = lane >> 4;
    const int k_begin = split * p.keys_per_split;
    const int k_end = min(p.S, k_begin + p.keys_per_split);
    const int nblk = (k_end - k_begin + KB - 1) / KB;
    const bf16* kb = p.k + b * p.k_bs + h * 32;
    const bf16* vb = p.v + b * p.v_bs + h * 32;
    const int NQ = nslices * MQ;
    const bool masked = p.mask != nullptr;
    // mask staging: dword d of the tile = (query d >> 4, keys 4 (d & 15) .. + 3); MREG dwords per thread cover 160 queries, more slices loop
    constexpr int MREG = MQ * 16 / 256;
    unsigned mreg[MREG];
    auto mload = [&](int blk, int pass) {
#pragma unroll
        for (int i = 0; i < MREG; ++i) {
            const int d = pass * MQ * 16 + threadIdx.x + 256 * i, q = d >> 4, kk = k_begin + blk * KB + 4 * (d & 15);
            mreg[i] = 0x01010101u;                      // padding queries / keys past the end: blocked
            if (q < p.Lq && kk < p.S) mreg[i] = *(const unsigned*)(p.mask + ((long)b * p.Lq + q) * p.mask_rs + kk);
        }
    };
    auto mstore = [&](int pass) {
#pragma unroll
        for (int i = 0; i < MREG; ++i) {
            const int d = pass * MQ * 16 + threadIdx.x + 256 * i, q = d >> 4, c4 = 4 * (d & 15);
#pragma unroll
            for (int j = 0; j < 4; ++j) Ms[(c4 + j) * MSTR + q] = (unsigned char)((mreg[i] >> (8 * j)) & 0xffu);
        }
    };

    stage_rows(Qs, p.q + b * p.q_bs + h * 32, p.q_rs, 0, NQ, p.Lq, 256);
    stage_rows(dOs, p.dout + b * p.do_bs + h * 32, p.do_rs, 0, NQ, p.Lq, 256);
    for (int qi = threadIdx.x; qi < NQ; qi += 256) {
        float l = 0.f, dsum = 0.f;
        if (qi < p.Lq) {
            l = p.lse[((long)b * p.H + h) * p.Lq + qi];
            const bf16* orow = p.out + b * p.o_bs + (long)qi * p.o_rs + h * 32;
            const bf16* grow = p.dout + b * p.do_bs + (long)qi * p.do_rs + h * 32;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const bf16x8 ov = *(const bf16x8*)(orow + c * 8), gv = *(const bf16x8*)(grow + c * 8);
#pragma unroll
                for (int j = 0; j < 8; ++j) dsum += (float)ov[j] * (float)gv[j];
            }
        }
        lse[qi] = l; delta[qi] = dsum;
    }
    const int srow = threadIdx.x >> 2, sch = threadIdx.x & 3;
    u32x4 rk, rv;
    auto gload = [&](int blk) {
        const int key = k_begin + blk * KB + srow;
        rk = (u32x4){0u, 0u, 0u, 0u}; rv = rk;
        if (key < k_end) {
            rk = *(const u32x4*)(kb + (long)key * p.k_rs + sch * 8);
            rv = *(const u32x4*)(vb + (long)key * p.v_rs + sch * 8);
        }
    };
    auto lstore = [&](int st) {
        *(u32x4*)(Ks + st * KB * 64 + rm_off(srow, sch)) = rk;
        *(u32x4*)(Vs + st * KB * 64 + rm_off(srow, sch)) = rv;
    };
    if (nblk > 0) { gload(0); lstore(0); }
    if (masked && nblk > 0)
        for (int pass = 0; pass < nslices; ++pass) { mload(0, pass); mstore(pass); }
    __syncthreads();

    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    const float sc = p.scale * LOG2E;
    for (int blk = 0; blk < nblk; ++blk) {
        const int st = blk & 1;
        if (blk + 1 < nblk) gload(blk + 1);
        if (masked && blk + 1 < nblk && nslices == 1) mload(blk + 1, 0);       // (one slice -- every shipped configuration: the next tile waits in registers)
        const unsigned char* Kc = Ks + st * KB * 64;
        const unsigned char* Vc = Vs + st * KB * 64;
        const int key = k_begin + blk * KB + wave * 16 + fr;           // this lane's key (column)
        const bf16x8 kfB = frag_rows(Kc, wave * 16, fr, fg);
        const bf16x8 vfB = frag_rows(Vc, wave * 16, fr, fg);
        f32x4 dk[2] = {zero4, zero4}, dv[2] = {zero4, zero4};
        for (int qb2 = 0; qb2 < nslices * NQB; ++qb2) {
            f32x4 pt[2], dst[2];
#pragma unroll
            for (int h2 = 0; h2 < 2; ++h2) {
                const int qt = 2 * qb2 + h2;
                const f32x4 sv = mfma16(frag_rows(Qs, qt * 16, fr, fg), kfB, zero4);       // S[q = 4fg + r][key = fr]
                const f32x4 dp = mfma16(frag_rows(dOs, qt * 16, fr, fg), vfB, zero4);
                const float4 l4 = *(const float4*)(lse + qt * 16 + 4 * fg);
                const float4 d4 = *(const float4*)(delta + qt * 16 + 4 * fg);
                const float lv[4] = {l4.x, l4.y, l4.z, l4.w}, dv4[4] = {d4.x, d4.y, d4.z, d4.w};
                unsigned mw = 0u;
                if (masked) mw = *(const unsigned*)(Ms + (wave * 16 + fr) * MSTR + qt * 16 + 4 * fg);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int qi = qt * 16 + 4 * fg + r;
                    // (no validity tests: a padding query has zero Q / dO rows, lse = delta = 0 -> adds nothing; a key past the end only
                    // feeds its own column, which is never stored; a blocked score becomes -inf -> p = 0)
                    float svr = sv[r];
                    if (masked) svr = (((mw >> (8 * r)) & 0xffu) == 0u) ? svr : NEG_INF;
                    const float pr = fast_exp2(__builtin_fmaf(svr, sc, -lv[r]));
                    float keepw = 1.0f;
                    const bool ok = DROP && (key < k_end) && (qi < p.Lq) && pr != 0.f;
                    if (DROP && ok)
                        keepw = attn_keep(p.seed, p.drop_thresh, ((unsigned long long)bh * p.Lq + qi) * p.S + key) ? p.inv_keep : 0.f;
                    pt[h2][r] = pr * keepw;
                    dst[h2][r] = pr * (dp[r] * keepw - dv4[r]);
                }
            }
            bf16x8 pb, db;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                pb[r] = (bf16)pt[0][r]; pb[4 + r] = (bf16)pt[1][r];
                db[r] = (bf16)dst[0][r]; db[4 + r] = (bf16)dst[1][r];
            }
#pragma unroll
            for (int dt = 0; dt < 2; ++dt) {
                dv[dt] = mfma16(frag_tr(dOs, 32 * qb2, 32 * qb2 + 16, dt * 16, lane), pb, dv[dt]);
                dk[dt] = mfma16(frag_tr(Qs, 32 * qb2, 32 * qb2 + 16, dt * 16, lane), db, dk[dt]);
            }
        }
        if (key < k_end) {
#pragma unroll
            for (int dt = 0; dt < 2; ++dt) {
                bf16x4 kv, vv;
#pragma unroll
                for (int r = 0; r < 4; ++r) { kv[r] = (bf16)(dk[dt][r] * p.scale); vv[r] = (bf16)dv[dt][r]; }
                *(bf16x4*)(p.dk + b * p.dk_bs + (long)key * p.dk_rs + h * 32 + dt * 16 + 4 * fg) = kv;
                *(bf16x4*)(p.dv + b * p.dv_bs + (long)key * p.dv_rs + h * 32 + dt * 16 + 4 * fg) = vv;
            }
        }
        if (blk + 1 < nblk) lstore(st ^ 1);
        if (masked && blk + 1 < nblk) {
            __syncthreads();                            // every wave is done with this block's mask tile
            if (nslices == 1) mstore(0);
            else for (int pass = 0; pass < nslices; ++pass) { mload(blk + 1, pass); mstore(pass); }
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------------
// host
// ------------------------------------------------------------------------------------------------
static int mha_splits(int B, int H, int S, int* keys_per_split) {
    const int nblocks = (S + KB - 1) / KB;
    int ns = (768 + B * H - 1) / (B * H);                 // ~3 workgroups per CU
    if (ns > nblocks) ns = nblocks;
    if (ns > 64) ns = 64;
    if (ns < 1) ns = 1;
    const int per = ((nblocks + ns - 1) / ns) * KB;
    *keys_per_split = per;
    return (S + per - 1) / per;
}

// workspace floats needed by the forward: nsplit partials of (m, l, O[32]) per (batch, head, query slice of 160)
extern "C" long uenc_mha_fwd_workspace_floats(int B, int H, int Lq, int S) {
    int per;
    const int ns = mha_splits(B, H, S, &per);
    if (ns == 1) return 0;
    const long slices = (Lq + MQ - 1) / MQ;
    return (long)B * H * slices * ns * MQ * 34;
}

static int mha_fill(MhaP& p, const void* q, long q_bs, long q_rs, const void* k, long k_bs, long k_rs, const void* v, long v_bs,
                    long v_rs, const unsigned char* mask, long mask_rs, int B, int H, int Lq, int S, float scale, float dropout_p,
                    unsigned seed) {
    if (!(dropout_p >= 0.f && dropout_p < 1.f)) return UENC_EINVAL;
    p.drop_thresh = attn_drop_thresh(dropout_p); p.seed = seed; p.inv_keep = 1.0f / (1.0f - dropout_p);
    if (!(q && k && v && B > 0 && H > 0 && Lq > 0 && S > 0)) return UENC_EINVAL;
    if ((q_rs | k_rs | v_rs | q_bs | k_bs | v_bs) & 7) return UENC_EINVAL;                   // 16-byte rows
    if (((uintptr_t)q | (uintptr_t)k | (uintptr_t)v) & 15) return UENC_EINVAL;
    if (mask != nullptr && (mask_rs % 4 != 0 || mask_rs < (S + 3) / 4 * 4 || ((uintptr_t)mask & 3))) return UENC_EINVAL;
    p.q = (const bf16*)q; p.k = (const bf16*)k; p.v = (const bf16*)v;
    p.q_bs = q_bs; p.q_rs = q_rs; p.k_bs = k_bs; p.k_rs = k_rs; p.v_bs = v_bs; p.v_rs = v_rs;
    p.mask = mask; p.mask_rs = mask_rs; p.B = B; p.H = H; p.Lq = Lq; p.S = S; p.scale = scale;
    p.nsplit = mha_splits(B, H, S, &p.keys_per_split);
    p.out = nullptr; p.lse = nullptr; p.ws_m = p.ws_l = p.ws_o = nullptr; p.dout = nullptr; p.dq = nullptr; p.dk = p.dv = nullptr;
    return UENC_OK;
}

// dropout_p / seed: dropout on the attention probabilities (training mode of nn.MultiheadAttention(dropout=p)); 0 = off.
// out (B, Lq, H*32-wide rows) bf16; lse (B, H, Lq) fp32 (log2 domain, needed by the backward; may be NULL);
// workspace: uenc_mha_fwd_workspace_floats() floats (may be NULL when that is 0).
extern "C" int uenc_mha_fwd(const void* q, long q_bs, long q_rs, const void* k, long k_bs, long k_rs, const void* v, long v_bs,
                            long v_rs, const unsigned char* mask, long mask_rs, void* out, long o_bs, long o_rs, float* lse, float* workspace,
                            int B, int H, int Lq, int S, float scale, float dropout_p, unsigned seed, hipStream_t stream) {
    MhaP p;
    int rc = mha_fill(p, q, q_bs, q_rs, k, k_bs, k_rs, v, v_bs, v_rs, mask, mask_rs, B, H, Lq, S, scale, dropout_p, seed);
    if (rc != UENC_OK) return rc;
    UENC_CHECK_ARG(out && (o_rs % 4 == 0) && (o_bs % 4 == 0));
    p.out = (bf16*)out; p.o_bs = o_bs; p.o_rs = o_rs; p.lse = lse;
    const int slices = (Lq + MQ - 1) / MQ;
    if (p.nsplit > 1) {
        UENC_CHECK_ARG(workspace != nullptr);
        const long parts = (long)B * H * slices * p.nsplit;
        p.ws_m = workspace; p.ws_l = workspace + parts * MQ; p.ws_o = workspace + 2 * parts * MQ;
    }
    if (p.drop_thresh != 0u) hipLaunchKernelGGL((mha_q_kernel<0, true>), dim3((unsigned)((p.nsplit * B * H + 15) / 16 * 16), 1, slices), dim3(256), 0, stream, p);
    else hipLaunchKernelGGL((mha_q_kernel<0, false>), dim3((unsigned)((p.nsplit * B * H + 15) / 16 * 16), 1, slices), dim3(256), 0, stream, p);
    if (p.nsplit > 1) {
        const long threads = (long)B * H * slices * MQ * 32;
        hipLaunchKernelGGL(mha_combine_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, stream, p, slices);
    }
    UENC_LAUNCH_RET();
}

// dq (B, Lq, *) fp32 accumulated (caller zeroes); dk, dv (B, S, *) bf16 overwritten for every key.
extern "C" int uenc_mha_bwd(const void* q, long q_bs, long q_rs, const void* k, long k_bs, long k_rs, const void* v, long v_bs,
                            long v_rs, const unsigned char* mask, long mask_rs, const void* out, long o_bs, long o_rs, const float* lse,
                            const void* dout, long do_bs, long do_rs, float* dq, long dq_bs, long dq_rs, void* dk, long dk_bs,
                            long dk_rs, void* dv, long dv_bs, long dv_rs, int B, int H, int Lq, int S, float scale, float dropout_p,
                            unsigned seed, hipStream_t stream) {
    MhaP p;
    int rc = mha_fill(p, q, q_bs, q_rs, k, k_bs, k_rs, v, v_bs, v_rs, mask, mask_rs, B, H, Lq, S, scale, dropout_p, seed);
    if (rc != UENC_OK) return rc;
    UENC_CHECK_ARG(out && lse && dout && dq && dk && dv);
    UENC_CHECK_ARG(((o_rs | do_rs | o_bs | do_bs) & 7) == 0 && ((dk_rs | dv_rs | dk_bs | dv_bs) & 3) == 0);
    p.out = (bf16*)out; p.o_bs = o_bs; p.o_rs = o_rs; p.lse = (float*)lse;
    p.dout = (const bf16*)dout; p.do_bs = do_bs; p.do_rs = do_rs;
    p.dq = dq; p.dq_bs = dq_bs; p.dq_rs = dq_rs;
    p.dk = (bf16*)dk; p.dk_bs = dk_bs; p.dk_rs = dk_rs; p.dv = (bf16*)dv; p.dv_bs = dv_bs; p.dv_rs = dv_rs;
    const int slices = (Lq + MQ - 1) / MQ;
    if (p.drop_thresh != 0u) hipLaunchKernelGGL((mha_q_kernel<1, true>), dim3((unsigned)((p.nsplit * B * H + 15) / 16 * 16), 1, slices), dim3(256), 0, stream, p);
    else hipLaunchKernelGGL((mha_q_kernel<1, false>), dim3((unsigned)((p.nsplit * B * H + 15) / 16 * 16), 1, slices), dim3(256), 0, stream, p);
    const size_t shm = (size_t)slices * MQ * 64 * 2 + 4 * KB * 64 + (size_t)slices * MQ * 8 + (mask != nullptr ? (size_t)KB * (slices * MQ + 4) : 0);
    if (shm > 160 * 1024) return UENC_EINVAL;
    const bool drop = p.drop_thresh != 0u;
    if (shm > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(drop ? (const void*)mha_dkdv_kernel<true> : (const void*)mha_dkdv_kernel<false>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
        if (e != hipSuccess) return (int)e;
    }
    if (drop) hipLaunchKernelGGL(mha_dkdv_kernel<true>, dim3((unsigned)((p.nsplit * B * H + 15) / 16 * 16)), dim3(256), shm, stream, p);
    else hipLaunchKernelGGL(mha_dkdv_kernel<false>, dim3((unsigned)((p.nsplit * B * H + 15) / 16 * 16)), dim3(256), shm, stream, p);
    UENC_LAUNCH_RET();
}
