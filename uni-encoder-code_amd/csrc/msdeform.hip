// Multi-scale deformable attention, forward and backward (gfx950).
//
// Drop-in for the reference's only native op, the pybind11 module `MultiScaleDeformableAttention`
// (model/modeling/pixel_decoder/ops/src/vision.cpp:18-21, ms_deform_attn.h:25-66; kernels
// src/cuda/ms_deform_im2col_cuda.cuh:242-304 forward, :306-408 backward for 32 channels/head):
//   out[b,q,m,:] = sum_{l,p} w[b,q,m,l,p] * bilinear(value_l[b,:,m,:], loc[b,q,m,l,p] * (W_l,H_l) - 0.5)
// with zero padding outside each level and per-tap bounds checks (cuh:38-89).
//
// HBM / gather bound: per (b,q,m) 4 taps x L*P samples x D channels.  Forward: LPG = D/4 lanes per
// (b,q,m), 16-byte tap loads (a tap's D channels are contiguous: 128 B for D=32 fp32, one line).
// Backward: one lane per channel; grad_loc / grad_attn are reduced over channels with wave shuffles (no LDS, no
// atomics).  grad_value: direct form = one atomic wave-instruction per tap covering two 128-byte row segments (the
// full-rate atomic shape); binned form (below) = records replayed per block of value pixels, see "backward, binned".
// `value` may be fp32 (the reference's contract) or bf16 (half the gather bytes).
#include "common.h"
#include <stdlib.h>

struct MsdaP {
    const void* value; int v_f32;
    const int64_t* shapes;       // (L, 2) = (H_l, W_l)
    const int64_t* level_start;  // (L)
    const float* loc;            // (B, Lq, M, L, P, 2)  (x, y) in [0, 1]
    const float* attn;           // (B, Lq, M, L, P)
    void* out; int out_f32;      // (B, Lq, M*D)
    const void* grad_out; int go_f32;
    float* grad_value;           // (B, S, M, D) accumulated (caller zeroes)
    float* grad_loc;             // (B, Lq, M, L, P, 2)
    float* grad_attn;            // (B, Lq, M, L, P)
    int B, S, M, D, L, Lq, P;
    // fused form (uenc_msdeform_attn_fused_*): locations and weights are derived in the kernel from the projection row
    //   offaw (B * Lq, ld) fp32 = [M][L][P][2] sampling offsets | [M][L * P] attention logits,  ref (B|1, Lq, L, 2):
    //   loc = ref + off / (W_l, H_l), attn = softmax over the L * P logits (ops/modules/ms_deform_attn.py:101-113)
    const float* offaw; long ld;
    const float* ref; int ref_per_image;
    bf16* doffaw; long ldd;      // backward: d(offaw) (B * Lq, ldd) bf16, every column of the 3 M L P written
};

__device__ __forceinline__ float4 ldv4(const void* base, int is_f32, long idx) {
    if (is_f32) return *(const float4*)((const float*)base + idx);
    const bf16x4 v = *(const bf16x4*)((const bf16*)base + idx);
    return make_float4((float)v[0], (float)v[1], (float)v[2], (float)v[3]);
}

// one bilinear sample (x, y in [0, 1], weight w) of level (Hl, Wl) added to acc: 4 channels of this lane
__device__ __forceinline__ void msda_fwd_sample(const MsdaP& p, float x, float y, float w, int Hl, int Wl, long lbase, long vstride, float4& acc) {
    const float him = y * Hl - 0.5f, wim = x * Wl - 0.5f;
    if (him > -1.f && wim > -1.f && him < (float)Hl && wim < (float)Wl) {
        const int h0 = (int)floorf(him), w0 = (int)floorf(wim);
        const float lh = him - h0, lw = wim - w0, hh = 1.f - lh, hw = 1.f - lw;
        const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
        float4 v1 = z, v2 = z, v3 = z, v4 = z;
        const long r0 = lbase + ((long)h0 * Wl + w0) * vstride;
        if (h0 >= 0 && w0 >= 0) v1 = ldv4(p.value, p.v_f32, r0);
        if (h0 >= 0 && w0 + 1 <= Wl - 1) v2 = ldv4(p.value, p.v_f32, r0 + vstride);
        if (h0 + 1 <= Hl - 1 && w0 >= 0) v3 = ldv4(p.value, p.v_f32, r0 + (long)Wl * vstride);
        if (h0 + 1 <= Hl - 1 && w0 + 1 <= Wl - 1) v4 = ldv4(p.value, p.v_f32, r0 + (long)Wl * vstride + vstride);
        const float w1 = hh * hw * w, w2 = hh * lw * w, w3 = lh * hw * w, w4 = lh * lw * w;
        acc.x += w1 * v1.x + w2 * v2.x + w3 * v3.x + w4 * v4.x;
        acc.y += w1 * v1.y + w2 * v2.y + w3 * v3.y + w4 * v4.y;
        acc.z += w1 * v1.z + w2 * v2.z + w3 * v3.z + w4 * v4.z;
        acc.w += w1 * v1.w + w2 * v2.w + w3 * v3.w + w4 * v4.w;
    }
}

template <int LPG, bool FUSED>
__global__ __launch_bounds__(256) void msda_fwd_kernel(MsdaP p) {
    const long gtid = (long)blockIdx.x * 256 + threadIdx.x;
    const long grp = gtid / LPG;
    const int c4 = (int)(gtid - grp * LPG) * 4;
    const long ngrp = (long)p.B * p.Lq * p.M;
    if (grp >= ngrp) return;
    const int m = (int)(grp % p.M);
    const long bq = grp / p.M;
    const int b = (int)(bq / p.Lq);
    const int LP = p.L * p.P;
    const long vstride = (long)p.M * p.D;               // between spatial positions
    const long vbase = (long)b * p.S * vstride + (long)m * p.D + c4;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    if (FUSED) {
        // softmax statistics first, then every sample's weight and location from the projection row as it is needed (L1-resident:
        // the group's 8 lanes read the same 144 bytes) -- nothing is held in register arrays, the kernel keeps its occupancy
        const float* off = p.offaw + bq * p.ld + (long)m * LP * 2;
        const float* lg = p.offaw + bq * p.ld + (long)p.M * LP * 2 + (long)m * LP;
        const float* rf = p.ref + (p.ref_per_image ? bq : bq % p.Lq) * p.L * 2;
        float mx = -3.0e38f, sum = 0.f;
        for (int j = 0; j < LP; ++j) mx = fmaxf(mx, lg[j]);
        for (int j = 0; j < LP; ++j) sum += __expf(lg[j] - mx);
        const float inv = 1.0f / sum;
        for (int l = 0; l < p.L; ++l) {
            const int Hl = (int)p.shapes[2 * l], Wl = (int)p.shapes[2 * l + 1];
            const long lbase = vbase + p.level_start[l] * vstride;
            const float rx = rf[2 * l], ry = rf[2 * l + 1];
            for (int k = 0; k < p.P; ++k) {
                const int j = l * p.P + k;
                const float2 o = *(const float2*)(off + 2 * j);
                msda_fwd_sample(p, rx + o.x / (float)Wl, ry + o.y / (float)Hl, __expf(lg[j] - mx) * inv, Hl, Wl, lbase, vstride, acc);
            }
        }
    } else {
        const float* loc = p.loc + grp * LP * 2;
        const float* aw = p.attn + grp * LP;
        for (int l = 0; l < p.L; ++l) {
            const int Hl = (int)p.shapes[2 * l], Wl = (int)p.shapes[2 * l + 1];
            const long lbase = vbase + p.level_start[l] * vstride;
            for (int k = 0; k < p.P; ++k)
                msda_fwd_sample(p, loc[(l * p.P + k) * 2], loc[(l * p.P + k) * 2 + 1], aw[l * p.P + k], Hl, Wl, lbase, vstride, acc);
        }
    }
    const long o = grp * p.D + c4;
    if (p.out_f32) *(float4*)((float*)p.out + o) = acc;
    else {
        bf16x4 ov; ov[0] = (bf16)acc.x; ov[1] = (bf16)acc.y; ov[2] = (bf16)acc.z; ov[3] = (bf16)acc.w;
        *(bf16x4*)((bf16*)p.out + o) = ov;
    }
}

// ---- backward, direct: D lanes per (b,q,m) (D = 16, 32 or 64), lane = channel, every tap one float atomic -------------
struct MsdaTap {       // geometry of one sample on its level
    int h0, w0;
    float lh, lw, hh, hw;
    bool y0ok, y1ok, x0ok, x1ok;
};
__device__ __forceinline__ bool msda_tap(float x, float y, int Hl, int Wl, MsdaTap& t) {
    const float him = y * Hl - 0.5f, wim = x * Wl - 0.5f;
    if (!(him > -1.f && wim > -1.f && him < (float)Hl && wim < (float)Wl)) return false;
    t.h0 = (int)floorf(him); t.w0 = (int)floorf(wim);
    t.lh = him - t.h0; t.lw = wim - t.w0; t.hh = 1.f - t.lh; t.hw = 1.f - t.lw;
    t.y0ok = t.h0 >= 0; t.y1ok = t.h0 + 1 <= Hl - 1; t.x0ok = t.w0 >= 0; t.x1ok = t.w0 + 1 <= Wl - 1;
    return true;
}

// The per-sample body shared by both forms: gathers the 4 taps, adds the taps selected by `direct` (bit dy*2+dx) to
// grad_value, and returns this lane's (channel's) share of d/dx, d/dy, d/dweight.
template <int D>
__device__ __forceinline__ void msda_sample_bwd(const MsdaP& p, const MsdaTap& t, int Hl, int Wl, long lbase, long vstride, float top,
                                                float w, unsigned direct, float& g_w, float& g_h, float& g_a) {
    const float tg = top * w;
    const long r0 = lbase + ((long)t.h0 * Wl + t.w0) * vstride;
    const long r1 = r0 + (long)Wl * vstride;
    float v1 = 0.f, v2 = 0.f, v3 = 0.f, v4 = 0.f;
    if (p.v_f32) {
        const float* vp = (const float*)p.value;
        if (t.y0ok && t.x0ok) v1 = vp[r0];
        if (t.y0ok && t.x1ok) v2 = vp[r0 + vstride];
        if (t.y1ok && t.x0ok) v3 = vp[r1];
        if (t.y1ok && t.x1ok) v4 = vp[r1 + vstride];
    } else {
        const bf16* vp = (const bf16*)p.value;
        if (t.y0ok && t.x0ok) v1 = (float)vp[r0];
        if (t.y0ok && t.x1ok) v2 = (float)vp[r0 + vstride];
        if (t.y1ok && t.x0ok) v3 = (float)vp[r1];
        if (t.y1ok && t.x1ok) v4 = (float)vp[r1 + vstride];
    }
    if (direct) {
        if ((direct & 1u) && t.y0ok && t.x0ok) atomicAdd(p.grad_value + r0, t.hh * t.hw * tg);
        if ((direct & 2u) && t.y0ok && t.x1ok) atomicAdd(p.grad_value + r0 + vstride, t.hh * t.lw * tg);
        if ((direct & 4u) && t.y1ok && t.x0ok) atomicAdd(p.grad_value + r1, t.lh * t.hw * tg);
        if ((direct & 8u) && t.y1ok && t.x1ok) atomicAdd(p.grad_value + r1 + vstride, t.lh * t.lw * tg);
    }
    g_a = top * (t.hh * t.hw * v1 + t.hh * t.lw * v2 + t.lh * t.hw * v3 + t.lh * t.lw * v4);
    g_w = (float)Wl * tg * (-t.hh * v1 + t.hh * v2 - t.lh * v3 + t.lh * v4);
    g_h = (float)Hl * tg * (-t.hw * v1 - t.lw * v2 + t.hw * v3 + t.lw * v4);
}

template <int D>
__device__ __forceinline__ void msda_store_sample_grads(const MsdaP& p, long slot, int c, float g_w, float g_h, float g_a) {
#pragma unroll
    for (int o = D / 2; o > 0; o >>= 1) {
        g_w += __shfl_xor(g_w, o);
        g_h += __shfl_xor(g_h, o);
        g_a += __shfl_xor(g_a, o);
    }
    if (c == 0) {
        *(float2*)(p.grad_loc + slot * 2) = make_float2(g_w, g_h);
        p.grad_attn[slot] = g_a;
    }
}

template <int D>
__global__ __launch_bounds__(256) void msda_bwd_kernel(MsdaP p) {
    const long gtid = (long)blockIdx.x * 256 + threadIdx.x;
    const long grp = gtid / D;
    const int c = (int)(gtid - grp * D);
    const long ngrp = (long)p.B * p.Lq * p.M;
    if (grp >= ngrp) return;       // D divides 64: whole groups leave together
    const int m = (int)(grp % p.M);
    const int b = (int)(grp / p.M / p.Lq);
    const int LP = p.L * p.P;
    const float* loc = p.loc + grp * LP * 2;
    const float* aw = p.attn + grp * LP;
    const long vstride = (long)p.M * D;
    const long vbase = (long)b * p.S * vstride + (long)m * D + c;
    const float top = p.go_f32 ? ((const float*)p.grad_out)[grp * D + c] : (float)((const bf16*)p.grad_out)[grp * D + c];
    for (int l = 0; l < p.L; ++l) {
        const int Hl = (int)p.shapes[2 * l], Wl = (int)p.shapes[2 * l + 1];
        const long lbase = vbase + p.level_start[l] * vstride;
        for (int k = 0; k < p.P; ++k) {
            const int s = l * p.P + k;
            float g_w = 0.f, g_h = 0.f, g_a = 0.f;
            MsdaTap t;
            if (msda_tap(loc[s * 2], loc[s * 2 + 1], Hl, Wl, t)) msda_sample_bwd<D>(p, t, Hl, Wl, lbase, vstride, top, aw[s], 0xfu, g_w, g_h, g_a);
            msda_store_sample_grads<D>(p, grp * LP + s, c, g_w, g_h, g_a);
        }
    }
}

// ---- backward, binned ------------------------------------------------------------------------------------------------
// Every float atomic is a 64-byte transaction at the memory side, ~1.3 TB/s chip-wide whatever the launch shape
// (MI355X_MICROARCH.md "Global float atomics"): the direct kernel sends 4 taps x L*P samples x D channels x 4 B per
// (query, head) there -- 4.2 GB per encoder layer at 1024 x 2048, ~3 ms -- and LDS float atomics (ds_add_f32) measured
// slower still (~3 cycles per lane).  The binned form sums on chip without float atomics.  Every level is cut into
// 4 x 4 blocks of value pixels, one bin of records per (image, head, block):
//   pass A (msda_bwd_bin_kernel): the gather half of the backward (grad_loc, grad_attn), 8 lanes x 4 channels per
//     (query, head); instead of adding its 4 taps to grad_value, a sample appends ONE 24-byte record ({query, tap position} +
//     4 tap weights) to the bin of each block its taps touch (1 - 4).  A workgroup = 32 consecutive queries of one head:
//     its appends are first counted per bin in an LDS hash table, so a bin's counter sees one returning atomic per
//     workgroup, not per record (integer atomics are memory-side transactions too: per-record counters cost as much as
//     the float atomics saved);
//   pass B (msda_bin_reduce_kernel): one wave per (bin, slice of its records) sums the records as a 16-pixel x 32-channel
//     MFMA product (see there), then adds the block to grad_value once.  Dense (coarse) levels get several slices per bin
//     so every wave sums a few hundred records.
// A sample that finds its bin full adds its taps directly, as the direct kernel does, so the result does not depend on
// how the samples are distributed -- only the speed does.
#define MSDA_TL 4
#define MSDA_BS 4               // block edge (pixels)
#define MSDA_CNT_STRIDE 32      // ints between bin counters: one 128-byte line each (counters sharing a line serialise)
#define MSDA_HASH 256
// A record, stored as two arrays (one 8-byte and one 16-byte load in pass B):
//   hd = {query, pos}: pos = (row + 1) | (column + 1) << 8 of tap (0,0) relative to the block (-1 .. 3);
//   w[dy*2+dx] = bilinear weight x attention weight of the tap, 0 outside the level
struct MsdaBins {
    int nbx[MSDA_TL], boff[MSDA_TL], cap[MSDA_TL];       // blocks per row; first bin of the level; records per bin of the level
    int nsplit[MSDA_TL], woff[MSDA_TL];                  // pass-B slices per bin; first work item of the level
    long roff[MSDA_TL];                                  // first record slot of the level within a (image, head)
    long rtot;                                           // record slots per (image, head)
    int nblk, nwork;                                     // bins / pass-B work items per (image, head)
    int variant;                                         // timing experiments only (UENC_MSDA_VARIANT), 0 in production
    int* count;                                          // [B * M * nblk] x MSDA_CNT_STRIDE ints (zeroed by the launcher)
    int2* rec_hd;                                        // [B * M][rtot]
    float4* rec_w;                                       // [B * M][rtot]
};

__device__ __forceinline__ void atomic_add4(float* g, const float4& v) {
    atomicAdd(g, v.x); atomicAdd(g + 1, v.y); atomicAdd(g + 2, v.z); atomicAdd(g + 3, v.w);
}
__device__ __forceinline__ float dot4(const float4& a, const float4& b) { return a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w; }

template <bool FUSED>
__global__ __launch_bounds__(256) void msda_bwd_bin_kernel(MsdaP p, MsdaBins bn) {
    constexpr int D = 32;
    __shared__ int h_key[MSDA_HASH], h_cnt[MSDA_HASH], h_base[MSDA_HASH];
    __shared__ float s_loc[FUSED ? 32 * 32 : 1], s_aw[FUSED ? 32 * 16 : 1];     // FUSED: the workgroup's 32 groups' locations / weights
    const int tid = threadIdx.x, j8 = tid & 7, c4 = j8 * 4;
    const int nchunk = (p.Lq + 31) / 32;
    const int chunk = blockIdx.x % nchunk, bm = blockIdx.x / nchunk;
    const int m = bm % p.M, b = bm / p.M;
    const int q = chunk * 32 + (tid >> 3);
    const bool live = q < p.Lq;
    const long grp = ((long)b * p.Lq + (live ? q : 0)) * p.M + m;
    const int LP = p.L * p.P;
    const float* loc = FUSED ? s_loc + (tid >> 3) * 32 : p.loc + grp * LP * 2;
    const float* aw = FUSED ? s_aw + (tid >> 3) * 16 : p.attn + grp * LP;
    const long row = (long)b * p.Lq + (live ? q : 0);
    if (FUSED) {
        // lane j8 of a group derives samples j8 and j8 + 8 from the projection row (msda_prep_kernel's arithmetic); max and sum over
        // the group's 8 lanes
        const float* off = p.offaw + row * p.ld + (long)m * LP * 2;
        const float* lg = p.offaw + row * p.ld + (long)p.M * LP * 2 + (long)m * LP;
        const long qr = p.ref_per_image ? row : row % p.Lq;
        const float* rf = p.ref + qr * p.L * 2;
        float lv[2], mx = -3.0e38f;
#pragma unroll
        for (int r = 0; r < 2; ++r) { const int s = j8 + 8 * r; lv[r] = s < LP ? lg[s] : -3.0e38f; mx = fmaxf(mx, lv[r]); }
        mx = fmaxf(mx, __shfl_xor(mx, 1)); mx = fmaxf(mx, __shfl_xor(mx, 2)); mx = fmaxf(mx, __shfl_xor(mx, 4));
        float ev[2], sum = 0.f;
#pragma unroll
        for (int r = 0; r < 2; ++r) { const int s = j8 + 8 * r; ev[r] = s < LP ? __expf(lv[r] - mx) : 0.f; sum += ev[r]; }
        sum = group8_sum(sum);
        const float inv = 1.0f / sum;
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const int s = j8 + 8 * r;
            if (s < LP) {
                const int l = s / p.P;
                const float2 o = *(const float2*)(off + 2 * s);
                s_loc[(tid >> 3) * 32 + 2 * s] = rf[2 * l] + o.x / (float)p.shapes[2 * l + 1];
                s_loc[(tid >> 3) * 32 + 2 * s + 1] = rf[2 * l + 1] + o.y / (float)p.shapes[2 * l];
                s_aw[(tid >> 3) * 16 + s] = ev[r] * inv;
            }
        }
    }
    h_key[tid] = -1; h_cnt[tid] = 0;
    __syncthreads();

    // ---- append phase: lane j8 of a group handles samples j8 and j8 + 8, each with up to 4 record-owning taps ----
    // code: 0 = no record; bit 31 set: bit 30 = slot taken straight from the global counter (hash table full) in bits 0-29,
    // else hash slot << 16 | position within this workgroup's share of the bin
    unsigned code[2][4];
#pragma unroll
    for (int round = 0; round < 2; ++round) {
        const int s = j8 + round * 8;
#pragma unroll
        for (int tp = 0; tp < 4; ++tp) code[round][tp] = 0u;
        if (live && s < LP && bn.variant != 1) {
            const int l = s / p.P;
            const int Hl = (int)p.shapes[2 * l], Wl = (int)p.shapes[2 * l + 1];
            MsdaTap t;
            if (msda_tap(loc[s * 2], loc[s * 2 + 1], Hl, Wl, t)) {
#pragma unroll
                for (int tp = 0; tp < 4; ++tp) {
                    const int dy = tp >> 1, dx = tp & 1;
                    const int cy = t.h0 + dy, cx = t.w0 + dx;
                    // a tap owns the record of its block unless an in-range tap of the same block precedes it
                    const bool dup_y = dy == 1 && t.y0ok && ((t.h0 + 1) / MSDA_BS) == (t.h0 / MSDA_BS);
                    const bool dup_x = dx == 1 && t.x0ok && ((t.w0 + 1) / MSDA_BS) == (t.w0 / MSDA_BS);
                    if (cy >= 0 && cy < Hl && cx >= 0 && cx < Wl && !dup_y && !dup_x) {
                        const int bin = bm * bn.nblk + bn.boff[l] + (cy / MSDA_BS) * bn.nbx[l] + cx / MSDA_BS;
                        int h = (int)(((unsigned)bin * 2654435761u) >> 24);
                        unsigned cd = 0u;
                        for (int probe = 0; probe < MSDA_HASH; ++probe) {
                            const int prev = atomicCAS(&h_key[h], -1, bin);
                            if (prev == -1 || prev == bin) { cd = 0x80000000u | ((unsigned)h << 16) | (unsigned)atomicAdd(&h_cnt[h], 1); break; }
                            h = (h + 1) & (MSDA_HASH - 1);
                        }
                        if (cd == 0u) cd = 0xc0000000u | ((unsigned)atomicAdd(bn.count + (long)bin * MSDA_CNT_STRIDE, 1) & 0x3fffffffu);
                        code[round][tp] = cd;
                    }
                }
            }
        }
    }
    __syncthreads();
    if (h_key[tid] >= 0) h_base[tid] = atomicAdd(bn.count + (long)h_key[tid] * MSDA_CNT_STRIDE, h_cnt[tid]);
    __syncthreads();
    unsigned ovfbits = 0u;          // bit 4 * round + tp: that record did not fit
#pragma unroll
    for (int round = 0; round < 2; ++round) {
        const int s = j8 + round * 8;
        if ((code[round][0] | code[round][1] | code[round][2] | code[round][3]) != 0u) {
            const int l = s / p.P;
            const int Hl = (int)p.shapes[2 * l], Wl = (int)p.shapes[2 * l + 1];
            MsdaTap t;
            msda_tap(loc[s * 2], loc[s * 2 + 1], Hl, Wl, t);
            const float w = aw[s];
            float wt[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) wt[k] = ((k >> 1) ? t.lh : t.hh) * ((k & 1) ? t.lw : t.hw) * w;
#pragma unroll
            for (int tp = 0; tp < 4; ++tp) {
                const unsigned cd = code[round][tp];
                if (cd == 0u) continue;
                const int slot = (cd & 0x40000000u) ? (int)(cd & 0x3fffffffu) : h_base[(cd >> 16) & 0xffu] + (int)(cd & 0xffffu);
                if (slot >= bn.cap[l]) { ovfbits |= 1u << (4 * round + tp); continue; }
                const int cy = t.h0 + (tp >> 1), cx = t.w0 + (tp & 1);
                const int by0 = cy / MSDA_BS * MSDA_BS, bx0 = cx / MSDA_BS * MSDA_BS;
                float wk[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int y = t.h0 + (k >> 1), x = t.w0 + (k & 1);
                    wk[k] = (y >= 0 && y < Hl && x >= 0 && x < Wl) ? wt[k] : 0.f;          // taps outside the level carry nothing
                }
                const long binl = (long)(cy / MSDA_BS) * bn.nbx[l] + cx / MSDA_BS;
                const long ri = (long)bm * bn.rtot + bn.roff[l] + binl * bn.cap[l] + slot;
                bn.rec_hd[ri] = make_int2(q, (int)((unsigned)(t.h0 - by0 + 1) | ((unsigned)(t.w0 - bx0 + 1) << 8)));
                bn.rec_w[ri] = make_float4(wk[0], wk[1], wk[2], wk[3]);
            }
        }
    }
    const bool any_ovf = __ballot(ovfbits != 0u) != 0ull;       // wave-uniform: the shuffles below are skipped when nothing overflowed
    if (!live || bn.variant == 2) return;

    // ---- gather phase: lane = 4 channels ----
    const int gl0 = (tid & 63) & ~7;           // first lane of this group within the wave
    const long vstride = (long)p.M * D;
    const long vbase = (long)b * p.S * vstride + (long)m * D + c4;
    const float4 top = ldv4(p.grad_out, p.go_f32, grp * D + c4);
    float keep_w[2] = {0.f, 0.f}, keep_h[2] = {0.f, 0.f}, keep_a[2] = {0.f, 0.f};
    for (int l = 0; l < p.L; ++l) {
        const int Hl = (int)p.shapes[2 * l], Wl = (int)p.shapes[2 * l + 1];
        const long lbase = vbase + p.level_start[l] * vstride;
        for (int k = 0; k < p.P; ++k) {
            const int s = l * p.P + k;
            float g_w = 0.f, g_h = 0.f, g_a = 0.f;
            MsdaTap t;
            const bool ok = msda_tap(loc[s * 2], loc[s * 2 + 1], Hl, Wl, t);
            unsigned o4 = 0u;
            if (any_ovf) o4 = ((unsigned)__shfl((int)ovfbits, gl0 + (s & 7)) >> (4 * (s >> 3))) & 0xfu;     // owner taps whose bin was full
            if (ok) {
                const float w = aw[s];
                const long r0 = lbase + ((long)t.h0 * Wl + t.w0) * vstride;
                const long r1 = r0 + (long)Wl * vstride;
                const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
                float4 v1 = z, v2 = z, v3 = z, v4 = z;
                if (t.y0ok && t.x0ok) v1 = ldv4(p.value, p.v_f32, r0);
                if (t.y0ok && t.x1ok) v2 = ldv4(p.value, p.v_f32, r0 + vstride);
                if (t.y1ok && t.x0ok) v3 = ldv4(p.value, p.v_f32, r1);
                if (t.y1ok && t.x1ok) v4 = ldv4(p.value, p.v_f32, r1 + vstride);
                if (o4) {
                    const int oy1 = (t.y0ok && ((t.h0 + 1) / MSDA_BS) == (t.h0 / MSDA_BS)) ? 0 : 1;     // owner row / column of the second taps
                    const int ox1 = (t.x0ok && ((t.w0 + 1) / MSDA_BS) == (t.w0 / MSDA_BS)) ? 0 : 1;
                    const float4 tg = make_float4(top.x * w, top.y * w, top.z * w, top.w * w);
                    auto scaled = [&](float f) { return make_float4(f * tg.x, f * tg.y, f * tg.z, f * tg.w); };
                    if (((o4 >> 0) & 1u) && t.y0ok && t.x0ok) atomic_add4(p.grad_value + r0, scaled(t.hh * t.hw));
                    if (((o4 >> ox1) & 1u) && t.y0ok && t.x1ok) atomic_add4(p.grad_value + r0 + vstride, scaled(t.hh * t.lw));
                    if (((o4 >> (2 * oy1)) & 1u) && t.y1ok && t.x0ok) atomic_add4(p.grad_value + r1, scaled(t.lh * t.hw));
                    if (((o4 >> (2 * oy1 + ox1)) & 1u) && t.y1ok && t.x1ok) atomic_add4(p.grad_value + r1 + vstride, scaled(t.lh * t.lw));
                }
                const float d1 = dot4(top, v1), d2 = dot4(top, v2), d3 = dot4(top, v3), d4 = dot4(top, v4);
                g_a = t.hh * t.hw * d1 + t.hh * t.lw * d2 + t.lh * t.hw * d3 + t.lh * t.lw * d4;
                g_w = (float)Wl * w * (-t.hh * d1 + t.hh * d2 - t.lh * d3 + t.lh * d4);
                g_h = (float)Hl * w * (-t.hw * d1 - t.lw * d2 + t.hw * d3 + t.lw * d4);
            }
            g_w = group8_sum(g_w); g_h = group8_sum(g_h); g_a = group8_sum(g_a);
            // every lane of the group now holds the sums: lane (s & 7) keeps sample s, so that the group's L*P results leave as
            // contiguous 64-byte rows (8 lanes x float2) instead of one 8-byte store per sample from lane 0
            if (j8 == (s & 7)) {
                if (s < 8) { keep_w[0] = g_w; keep_h[0] = g_h; keep_a[0] = g_a; }
                else { keep_w[1] = g_w; keep_h[1] = g_h; keep_a[1] = g_a; }
            }
        }
    }
    if (FUSED) {
        // d(offaw) straight from the group's registers: d(offset) = d(loc) / (W_l, H_l), d(logit) = aw * (d(aw) - sum_s aw d(aw))
        // (the arithmetic of msda_prep_kernel<true>); lanes write consecutive 4- / 2-byte pieces of the row: coalesced runs
        float part = 0.f;
#pragma unroll
        for (int rnd = 0; rnd < 2; ++rnd) { const int s = j8 + 8 * rnd; if (s < LP) part += aw[s] * keep_a[rnd]; }
        const float dot = group8_sum(part);
        bf16* drow = p.doffaw + row * p.ldd;
#pragma unroll
        for (int rnd = 0; rnd < 2; ++rnd) {
            const int s = j8 + 8 * rnd;
            if (s < LP) {
                const int l = s / p.P;
                typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
                bf16x2 d2;
                d2[0] = (bf16)(keep_w[rnd] / (float)p.shapes[2 * l + 1]);
                d2[1] = (bf16)(keep_h[rnd] / (float)p.shapes[2 * l]);
                *(bf16x2*)(drow + (long)m * LP * 2 + 2 * s) = d2;
                drow[(long)p.M * LP * 2 + (long)m * LP + s] = (bf16)(aw[s] * (keep_a[rnd] - dot));
            }
        }
        return;
    }
#pragma unroll
    for (int rnd = 0; rnd < 2; ++rnd) {
        const int s = j8 + 8 * rnd;
        if (s < LP) {
            *(float2*)(p.grad_loc + (grp * LP + s) * 2) = make_float2(keep_w[rnd], keep_h[rnd]);
            p.grad_attn[grp * LP + s] = keep_a[rnd];
        }
    }
}

// pass B: one wave per (bin, slice of its records).  The block's gradient is an outer-product sum
//   G[pixel][channel] = sum_r W[pixel][r] * T[r][channel],   W = tap weight of record r at the pixel (4 non-zeros per record),
//                                                            T = grad_out row of the record's query,
// i.e. a 16 x 32 x (records) GEMM: two v_mfma_f32_16x16x32_bf16 chains (channels 0-15 / 16-31) with K = 32 records per step
// and the 16 pixels of the block as M.  fp32 operands enter as exact sums of bf16 terms (w = w1 + w2 + w3, 8 + 8 + 8 mantissa
// bits; a bf16 grad_out is one term already), every partial product is exact in the fp32 accumulator, so the result is an
// fp32-accumulated sum like the direct kernel's -- without LDS, float atomics or read-modify-write chains in the loop.
__device__ __forceinline__ bf16 bf16_bits(unsigned u) {
    const unsigned short h = (unsigned short)(u >> 16);
    return __builtin_bit_cast(bf16, h);
}
// v = t[0] + t[1] + t[2] exactly (truncating splits: each remainder is exactly representable)
struct Bf3 { bf16 t[3]; };
__device__ __forceinline__ Bf3 split3(float v) {
    const unsigned u0 = __float_as_uint(v) & 0xffff0000u;
    const float r1 = v - __uint_as_float(u0);
    const unsigned u1 = __float_as_uint(r1) & 0xffff0000u;
    const float r2 = r1 - __uint_as_float(u1);
    Bf3 o;
    o.t[0] = bf16_bits(u0); o.t[1] = bf16_bits(u1); o.t[2] = bf16_bits(__float_as_uint(r2));
    return o;
}

template <bool GO_F32>
__global__ __launch_bounds__(256) void msda_bin_reduce_kernel(MsdaP p, MsdaBins bn) {
    constexpr int D = 32, NT = GO_F32 ? 3 : 1;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63, n16 = lane & 15, g = lane >> 4;
    const long nitems = (long)p.B * p.M * bn.nwork;
    const long item = (long)blockIdx.x * 4 + wave;
    if (item >= nitems) return;
    const int bm = (int)(item / bn.nwork);
    const int r = (int)(item - (long)bm * bn.nwork);
    int l = 0;
    while (l + 1 < p.L && r >= bn.woff[l + 1]) ++l;
    const int ns = bn.nsplit[l];
    const int binl = (r - bn.woff[l]) / ns;
    const int k = (r - bn.woff[l]) - binl * ns;
    int n = bn.count[((long)bm * bn.nblk + bn.boff[l] + binl) * MSDA_CNT_STRIDE];
    n = n < bn.cap[l] ? n : bn.cap[l];
    const int per = (n + ns - 1) / ns;
    const int lo = k * per;
    const int cnt = (lo + per < n ? lo + per : n) - lo;
    if (cnt <= 0) return;
    const int b = bm / p.M, m = bm - b * p.M;
    const long rbase = (long)bm * bn.rtot + bn.roff[l] + (long)binl * bn.cap[l] + lo;
    const int2* rec_hd = bn.rec_hd + rbase;
    const float4* rec_w = bn.rec_w + rbase;
    const long gobase = (long)b * p.Lq * p.M * D + (long)m * D + 2 * n16;        // this lane's channel pair (2 n16, 2 n16 + 1)
    const long gostride = (long)p.M * D;
    const int py = n16 >> 2, px = n16 & 3;                 // this lane's pixel (row of the W operand)
    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
    for (int i0 = 0; i0 < cnt; i0 += 32) {
        bf16x8 wa[3], t0[NT], t1[NT];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int idx = i0 + 8 * g + j;
            int2 hd = make_int2(0, 0);
            float4 w4 = make_float4(0.f, 0.f, 0.f, 0.f);
            if (idx < cnt) { hd = rec_hd[idx]; w4 = rec_w[idx]; }
            const int dy = py - ((hd.y & 0xff) - 1), dx = px - (((hd.y >> 8) & 0xff) - 1);
            float wv = dx ? (dy ? w4.w : w4.y) : (dy ? w4.z : w4.x);
            if ((unsigned)dy > 1u || (unsigned)dx > 1u) wv = 0.f;
            { const Bf3 ws = split3(wv); wa[0][j] = ws.t[0]; wa[1][j] = ws.t[1]; wa[2][j] = ws.t[2]; }
            const long gi = gobase + (long)hd.x * gostride;
            if (GO_F32) {
                const float2 gv = *(const float2*)((const float*)p.grad_out + gi);
                const Bf3 s0 = split3(gv.x), s1 = split3(gv.y);
#pragma unroll
                for (int c = 0; c < NT; ++c) { t0[c][j] = s0.t[c]; t1[c][j] = s1.t[c]; }
            } else {
                const unsigned pr = *(const unsigned*)((const bf16*)p.grad_out + gi);          // two channels in one dword
                t0[0][j] = bf16_bits(pr << 16);
                t1[0][j] = bf16_bits(pr & 0xffff0000u);
            }
        }
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int c = 0; c < NT; ++c)
                if (a + c <= 2) {                          // terms below 2^-24 of the product are dropped
                    acc0 = mfma16(wa[a], t0[c], acc0);
                    acc1 = mfma16(wa[a], t1[c], acc1);
                }
    }
    // acc: lane holds pixels 4g .. 4g+3 (register index) of channel 2 n16 (acc0) and 2 n16 + 1 (acc1)
    const int Hl = (int)p.shapes[2 * l], Wl = (int)p.shapes[2 * l + 1];
    const int by0 = (binl / bn.nbx[l]) * MSDA_BS, bx0 = (binl % bn.nbx[l]) * MSDA_BS;
    const long vstride = (long)p.M * D;
    const long lbase = (long)b * p.S * vstride + (long)m * D + 2 * n16 + p.level_start[l] * vstride;
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
        const int y = by0 + g, x = bx0 + rr;               // pixel 4g + rr = (row g, column rr)
        if (y < Hl && x < Wl) {
            float* gp = p.grad_value + lbase + ((long)y * Wl + x) * vstride;
            if (acc0[rr] != 0.f) atomicAdd(gp, acc0[rr]);
            if (acc1[rr] != 0.f) atomicAdd(gp + 1, acc1[rr]);
        }
    }
}

// Bin geometry: per level, capacity = 2 x the record count of evenly spread samples (incl. the copies of block-straddling
// samples, (1 + 1/4)^2 per sample), at least 64; pass-B slices per bin so that a wave sums ~384 records.
static int msda_plan_bins(const int64_t* shapes_host, int L, int Lq, int P, int D, MsdaBins& bn) {
    if (L > MSDA_TL || D != 32 || L * P > 16) return 0;
    int off = 0, woff = 0;
    long roff = 0;
    for (int l = 0; l < MSDA_TL; ++l) {
        bn.nbx[l] = 1; bn.boff[l] = off; bn.cap[l] = 0; bn.roff[l] = roff; bn.nsplit[l] = 1; bn.woff[l] = woff;
        if (l >= L) continue;
        const long Hl = shapes_host[2 * l], Wl = shapes_host[2 * l + 1];
        if (Hl <= 0 || Wl <= 0) return 0;
        const long nby = (Hl + MSDA_BS - 1) / MSDA_BS, nbx = (Wl + MSDA_BS - 1) / MSDA_BS;
        const double expect = (double)Lq * P / (double)(nby * nbx) * 1.5625;
        long cap = (long)(2.0 * expect) + 1;
        cap = cap < 64 ? 64 : (cap + 31) / 32 * 32;
        if (cap > 0xffff0) return 0;
        long ns = (long)(expect / 384.0 + 0.5);
        ns = ns < 1 ? 1 : (ns > 64 ? 64 : ns);
        bn.nbx[l] = (int)nbx; bn.cap[l] = (int)cap; bn.nsplit[l] = (int)ns;
        off += (int)(nby * nbx);
        woff += (int)(nby * nbx * ns);
        roff += nby * nbx * cap;
    }
    bn.nblk = off;
    bn.nwork = woff;
    bn.rtot = roff;
    return 1;
}

static int msda_fill(MsdaP& p, const void* value, int v_dtype, const int64_t* shapes, const int64_t* level_start,
                     const float* loc, const float* attn, int B, int S, int M, int D, int L, int Lq, int P) {
    if (!(value && shapes && level_start && loc && attn)) return UENC_EINVAL;
    if (!(B > 0 && S > 0 && M > 0 && L > 0 && L <= 16 && Lq > 0 && P > 0 && P <= 16)) return UENC_EINVAL;
    if (!(v_dtype == UENC_F32 || v_dtype == UENC_BF16)) return UENC_EINVAL;
    if ((uintptr_t)value & 15) return UENC_EINVAL;
    p.value = value; p.v_f32 = (v_dtype == UENC_F32); p.shapes = shapes; p.level_start = level_start;
    p.loc = loc; p.attn = attn; p.B = B; p.S = S; p.M = M; p.D = D; p.L = L; p.Lq = Lq; p.P = P;
    p.out = nullptr; p.grad_out = nullptr; p.grad_value = nullptr; p.grad_loc = nullptr; p.grad_attn = nullptr;
    p.offaw = nullptr; p.ld = 0; p.ref = nullptr; p.ref_per_image = 0; p.doffaw = nullptr; p.ldd = 0;
    return UENC_OK;
}

// Mirrors ms_deform_attn_forward(value, spatial_shapes, level_start_index, sampling_loc, attn_weight, im2col_step)
// (im2col_step only batches launches in the reference; one launch covers the whole batch here).
extern "C" int uenc_msdeform_attn_fwd(const void* value, int v_dtype, const int64_t* shapes, const int64_t* level_start,
                                      const float* loc, const float* attn, void* out, int out_dtype, int B, int S, int M,
                                      int D, int L, int Lq, int P, hipStream_t stream) {
    MsdaP p;
    int rc = msda_fill(p, value, v_dtype, shapes, level_start, loc, attn, B, S, M, D, L, Lq, P);
    if (rc != UENC_OK) return rc;
    UENC_CHECK_ARG(out && (D == 16 || D == 32 || D == 64));
    p.out = out; p.out_f32 = (out_dtype == UENC_F32);
    const long threads = (long)B * Lq * M * (D / 4);
    const unsigned grid = (unsigned)((threads + 255) / 256);
    if (D == 32) hipLaunchKernelGGL((msda_fwd_kernel<8, false>), dim3(grid), dim3(256), 0, stream, p);
    else if (D == 16) hipLaunchKernelGGL((msda_fwd_kernel<4, false>), dim3(grid), dim3(256), 0, stream, p);
    else hipLaunchKernelGGL((msda_fwd_kernel<16, false>), dim3(grid), dim3(256), 0, stream, p);
    UENC_LAUNCH_RET();
}

// Bytes of scratch the binned backward needs (0: shape not eligible, the direct kernel runs and needs none).
extern "C" long uenc_msdeform_attn_bwd_workspace_bytes(const int64_t* shapes_host, int B, int M, int D, int L, int Lq, int P) {
    MsdaBins bn;
    if (!shapes_host || B <= 0 || M <= 0 || !msda_plan_bins(shapes_host, L, Lq, P, D, bn)) return 0;
    const long nbins = (long)B * M * bn.nblk;
    return nbins * MSDA_CNT_STRIDE * 4 + (long)B * M * bn.rtot * 24;
}

// Mirrors ms_deform_attn_backward(...): grad_value must be zero-filled by the caller (it is accumulated);
// grad_loc / grad_attn are fully overwritten.  With shapes_host (host copy of `shapes`) and a workspace of
// uenc_msdeform_attn_bwd_workspace_bytes() bytes the binned kernels run, else the direct-atomics one.
extern "C" int uenc_msdeform_attn_bwd(const void* value, int v_dtype, const int64_t* shapes, const int64_t* level_start,
                                      const float* loc, const float* attn, const void* grad_out, int go_dtype,
                                      float* grad_value, float* grad_loc, float* grad_attn, int B, int S, int M, int D, int L,
                                      int Lq, int P, const int64_t* shapes_host, void* workspace, long workspace_bytes,
                                      hipStream_t stream) {
    MsdaP p;
    int rc = msda_fill(p, value, v_dtype, shapes, level_start, loc, attn, B, S, M, D, L, Lq, P);
    if (rc != UENC_OK) return rc;
    UENC_CHECK_ARG(grad_out && grad_value && grad_loc && grad_attn && (D == 32 || D == 64 || D == 16));
    UENC_CHECK_ARG(((uintptr_t)grad_out & 15) == 0);
    p.grad_out = grad_out; p.go_f32 = (go_dtype == UENC_F32);
    p.grad_value = grad_value; p.grad_loc = grad_loc; p.grad_attn = grad_attn;
    MsdaBins bn;
    if (shapes_host != nullptr && workspace != nullptr && msda_plan_bins(shapes_host, L, Lq, P, D, bn)) {
        { const char* e = getenv("UENC_MSDA_VARIANT"); bn.variant = e ? atoi(e) : 0; }
        long tot = 0;
        for (int l = 0; l < L; ++l) tot += shapes_host[2 * l] * shapes_host[2 * l + 1];
        const long nbins = (long)B * M * bn.nblk;
        const long cnt_bytes = nbins * MSDA_CNT_STRIDE * 4;
        const long nchunk = (Lq + 31) / 32;
        UENC_CHECK_ARG(tot == S && ((uintptr_t)workspace & 15) == 0 &&
                       workspace_bytes >= cnt_bytes + (long)B * M * bn.rtot * 24);
        const long nitems = (long)B * M * bn.nwork;
        UENC_CHECK_ARG(nbins < (1L << 31) && nchunk * B * M < (1L << 31) && (nitems + 3) / 4 < (1L << 31));
        bn.count = (int*)workspace;
        bn.rec_w = (float4*)((char*)workspace + cnt_bytes);                     // 16-byte records first (alignment)
        bn.rec_hd = (int2*)((char*)workspace + cnt_bytes + (long)B * M * bn.rtot * 16);
        hipError_t e = hipMemsetAsync(bn.count, 0, (size_t)cnt_bytes, stream);
        if (e != hipSuccess) return (int)e;
        hipLaunchKernelGGL(msda_bwd_bin_kernel<false>, dim3((unsigned)(nchunk * B * M)), dim3(256), 0, stream, p, bn);
        if (p.go_f32) hipLaunchKernelGGL(msda_bin_reduce_kernel<true>, dim3((unsigned)((nitems + 3) / 4)), dim3(256), 0, stream, p, bn);
        else hipLaunchKernelGGL(msda_bin_reduce_kernel<false>, dim3((unsigned)((nitems + 3) / 4)), dim3(256), 0, stream, p, bn);
        UENC_LAUNCH_RET();
    }
    const long threads = (long)B * Lq * M * D;
    const unsigned grid = (unsigned)((threads + 255) / 256);
    if (D == 32) hipLaunchKernelGGL(msda_bwd_kernel<32>, dim3(grid), dim3(256), 0, stream, p);
    else if (D == 64) hipLaunchKernelGGL(msda_bwd_kernel<64>, dim3(grid), dim3(256), 0, stream, p);
    else hipLaunchKernelGGL(msda_bwd_kernel<16>, dim3(grid), dim3(256), 0, stream, p);
    UENC_LAUNCH_RET();
}


// ---- fused form: the sampling locations and attention weights never exist in memory ------------------------------------------
// What ops/modules/ms_deform_attn.py:99-125 does between its two projection Linears and the native op, inside the op: the row
// offaw = [sampling_offsets | attention_weights] projection output (one GEMM over the stacked weights), the reference points and the
// level shapes go in; softmax over the L * P logits and loc = ref + off / (W_l, H_l) happen in registers (forward) / LDS (backward),
// and the backward returns d(offaw) directly.  Saves, per encoder layer at 1024 x 2048, the 99 MB loc / attn round trip each way
// and the two glue kernels (uenc_msda_prep_*).  L * P <= 16, D = 32, ld % 2 == 0; the backward needs the binned plan
// (shapes_host + workspace as for uenc_msdeform_attn_bwd) and returns -1 where that plan does not exist.
static int msda_fill_fused(MsdaP& p, const void* value, int v_dtype, const int64_t* shapes, const int64_t* level_start, const float* offaw,
                           long ld, const float* ref, int ref_per_image, int B, int S, int M, int D, int L, int Lq, int P) {
    if (!(offaw && ref)) return UENC_EINVAL;
    int rc = msda_fill(p, value, v_dtype, shapes, level_start, offaw, offaw, B, S, M, D, L, Lq, P);
    if (rc != UENC_OK) return rc;
    if (!(L * P <= 16 && D == 32 && ld >= (long)3 * M * L * P && ld % 2 == 0 && ((uintptr_t)offaw & 7) == 0)) return UENC_EINVAL;
    p.loc = nullptr; p.attn = nullptr;
    p.offaw = offaw; p.ld = ld; p.ref = ref; p.ref_per_image = ref_per_image;
    return UENC_OK;
}

extern "C" int uenc_msdeform_attn_fused_fwd(const void* value, int v_dtype, const int64_t* shapes, const int64_t* level_start,
                                            const float* offaw, long ld, const float* ref, int ref_per_image, void* out, int out_dtype,
                                            int B, int S, int M, int D, int L, int Lq, int P, hipStream_t stream) {
    MsdaP p;
    int rc = msda_fill_fused(p, value, v_dtype, shapes, level_start, offaw, ld, ref, ref_per_image, B, S, M, D, L, Lq, P);
    if (rc != UENC_OK) return rc;
    UENC_CHECK_ARG(out != nullptr);
    p.out = out; p.out_f32 = (out_dtype == UENC_F32);
    const long threads = (long)B * Lq * M * (D / 4);
    hipLaunchKernelGGL((msda_fwd_kernel<8, true>), dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, stream, p);
    UENC_LAUNCH_RET();
}

extern "C" int uenc_msdeform_attn_fused_bwd(const void* value, int v_dtype, const int64_t* shapes, const int64_t* level_start,
                                            const float* offaw, long ld, const float* ref, int ref_per_image, const void* grad_out,
                                            int go_dtype, float* grad_value, void* doffaw, long ld_doffaw, int B, int S, int M, int D, int L, int Lq, int P,
                                            const int64_t* shapes_host, void* workspace, long workspace_bytes, hipStream_t stream) {
    MsdaP p;
    int rc = msda_fill_fused(p, value, v_dtype, shapes, level_start, offaw, ld, ref, ref_per_image, B, S, M, D, L, Lq, P);
    if (rc != UENC_OK) return rc;
    UENC_CHECK_ARG(grad_out && grad_value && doffaw && shapes_host && workspace);
    UENC_CHECK_ARG(((uintptr_t)grad_out & 15) == 0 && ((uintptr_t)doffaw & 3) == 0 && ld_doffaw >= (long)3 * M * L * P && ld_doffaw % 2 == 0);
    p.grad_out = grad_out; p.go_f32 = (go_dtype == UENC_F32); p.grad_value = grad_value; p.doffaw = (bf16*)doffaw; p.ldd = ld_doffaw;
    MsdaBins bn;
    if (!msda_plan_bins(shapes_host, L, Lq, P, D, bn)) return UENC_EINVAL;
    { const char* e = getenv("UENC_MSDA_VARIANT"); bn.variant = e ? atoi(e) : 0; }
    long tot = 0;
    for (int l = 0; l < L; ++l) tot += shapes_host[2 * l] * shapes_host[2 * l + 1];
    const long nbins = (long)B * M * bn.nblk;
    const long cnt_bytes = nbins * MSDA_CNT_STRIDE * 4;
    const long nchunk = (Lq + 31) / 32;
    UENC_CHECK_ARG(tot == S && ((uintptr_t)workspace & 15) == 0 && workspace_bytes >= cnt_bytes + (long)B * M * bn.rtot * 24);
    const long nitems = (long)B * M * bn.nwork;
    UENC_CHECK_ARG(nbins < (1L << 31) && nchunk * B * M < (1L << 31) && (nitems + 3) / 4 < (1L << 31));
    bn.count = (int*)workspace;
    bn.rec_w = (float4*)((char*)workspace + cnt_bytes);
    bn.rec_hd = (int2*)((char*)workspace + cnt_bytes + (long)B * M * bn.rtot * 16);
    hipError_t e = hipMemsetAsync(bn.count, 0, (size_t)cnt_bytes, stream);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(msda_bwd_bin_kernel<true>, dim3((unsigned)(nchunk * B * M)), dim3(256), 0, stream, p, bn);
    if (p.go_f32) hipLaunchKernelGGL(msda_bin_reduce_kernel<true>, dim3((unsigned)((nitems + 3) / 4)), dim3(256), 0, stream, p, bn);
    else hipLaunchKernelGGL(msda_bin_reduce_kernel<false>, dim3((unsigned)((nitems + 3) / 4)), dim3(256), 0, stream, p, bn);
    UENC_LAUNCH_RET();
}
