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

// ---- forward, value tiles in LDS (the deformable ENCODER: the queries are the pixels of the L maps themselves) ----------
// The gather kernel above reads every tap from L2: 4 taps x L*P samples x 64 bytes per (query, head) = 2.1 GB per encoder layer at
// 1024 x 2048, 284 us.  In the encoder the queries of a neighbourhood sample a neighbourhood: a workgroup takes ONE head and ONE
// region of the image -- a TY x TX rectangle of the finest level plus the pixels of the other levels that fall into the same fraction of
// the image -- and
//   1. finds, per level, the bounding box of the taps its queries really touch (from the sampling locations: nothing is assumed
//      about the offsets),
//   2. copies the boxes that fit the LDS budget into LDS (64 bytes per pixel of this head, each read from L2 once per workgroup),
//   3. gathers from LDS; a level whose box does not fit is gathered from global memory as before (so any input gives the
//      reference's result; only the speed depends on how far the offsets reach).
// A thread owns a (query, head) with all 32 channels.  LDS pixel images are 64 bytes; chunk c of pixel i sits at slot
// c ^ ((i >> 2) & 3), so that 16 lanes reading the same chunk of 16 neighbouring pixels cover all 64 banks.
#define MSDT_L 4
struct MsdaTile {
    int H[MSDT_L], W[MSDT_L], start[MSDT_L];
    int RH, RW;                 // the map the regions are cut from: the FINEST level (the pixel decoder lists its levels coarse to fine)
    int TY, TX, ntx, nregions;  // region size on that map, regions per row / per image
    int lds_bytes;              // budget for the value tiles
    int variant;                // timing experiments (UENC_MSDA_VARIANT): 1 = no staging, 2 = stop after staging; 0 in production
};

__device__ __forceinline__ void msdt_fma8(const u32x4& a, const u32x4& b, const u32x4& c, const u32x4& d, float w1, float w2, float w3, float w4, float* acc) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        acc[2 * i] += w1 * __uint_as_float(a[i] << 16) + w2 * __uint_as_float(b[i] << 16) + w3 * __uint_as_float(c[i] << 16) + w4 * __uint_as_float(d[i] << 16);
        acc[2 * i + 1] += w1 * __uint_as_float(a[i] & 0xffff0000u) + w2 * __uint_as_float(b[i] & 0xffff0000u) + w3 * __uint_as_float(c[i] & 0xffff0000u) +
                          w4 * __uint_as_float(d[i] & 0xffff0000u);
    }
}

// P4: P == 4 -- a level's four locations / weights are three 16-byte loads, and all levels' loads are requested before the first sample is
// used (with a run-time P every sample's location was a dependent global load in front of its LDS reads: 12 exposed L2 round trips per
// query, and the kernel ran 1.5 x SLOWER than the gather kernel it replaces).
// BWD: the gather half of the backward in the fused form (uenc_msdeform_attn_fused_bwd; P == 4): locations and softmaxed weights are
// derived from the projection row `offaw` as in msda_bwd_bin_kernel<true>, the thread dots the 32 channels of its query's grad_out row
// with the four taps of every sample and writes d(offaw) -- the grad_value half stays with the binned kernels.
__device__ __forceinline__ void msdt_dot8(const u32x4& v, const float* t8, float& d) {
#pragma unroll
    for (int i = 0; i < 4; ++i) d += t8[2 * i] * __uint_as_float(v[i] << 16) + t8[2 * i + 1] * __uint_as_float(v[i] & 0xffff0000u);
}

// (the backward form keeps 32 grad_out channels, the samples' results and the tap reads live: ~220 registers, so it runs 256-thread
// workgroups at two waves per SIMD on 8 x 24 regions -- 252 queries -- where the forward runs 384 threads at three on 8 x 32)
// MODE 0: forward from sampling locations / attention weights; 1: the backward above; 2: forward from the projection row (no msda_prep pass,
// no loc / attn tensors: the fused forward the backward has had all along).
template <bool P4, int MODE>
__global__ __launch_bounds__(MODE == 1 ? 256 : 384, MODE == 1 ? 2 : 3) void msda_tiled_kernel(MsdaP p, MsdaTile g) {
    constexpr bool BWD = MODE == 1, FUSED_IN = MODE != 0;
    static_assert(P4 || !FUSED_IN, "the fused forms are built for P == 4");
    extern __shared__ __attribute__((aligned(16))) unsigned char vt[];
    __shared__ int s_box[MSDT_L][4];                     // ymin, ymax, xmin, xmax of the taps on each level
    __shared__ int s_part[6][MSDT_L][4];                 // ... per wave
    __shared__ int s_lv[MSDT_L][8];                      // box origin (y, x), box width, LDS offset (-1: not staged), H, W, first pixel: read back per level in pass 2
    const int tid = threadIdx.x, nth = blockDim.x, lane = tid & 63, wave = tid >> 6, nwave = nth >> 6;
    // workgroup -> (image, region, head): the M heads of a region are CONSECUTIVE workgroups of ONE XCD (hardware block i runs on XCD i % 8).
    // Head m uses 96 of the 768 bytes of a query's sampling-location row and 64 of the 512 bytes of a value pixel: dispatched 128 workgroups
    // apart on eight different XCDs, every head re-fetched the same lines from the fabric (pass 1 alone took 230 us of a 350 us launch).
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int m = slot % p.M, unit = (slot / p.M) * 8 + xcd;
    const int b = unit / g.nregions, region = unit - b * g.nregions;
    if (b >= p.B) return;
    if (g.variant & 4) return;
    const int ry = region / g.ntx, rx = region - ry * g.ntx;
    const int LP = p.L * p.P;
    // this region's rectangle of queries on every level: [ceil(t0 * H_l / RH), ceil(t1 * H_l / RH)) partitions level l exactly
    const int ty0 = ry * g.TY, ty1 = min(ty0 + g.TY, g.RH), tx0 = rx * g.TX, tx1 = min(tx0 + g.TX, g.RW);
    int qy0[MSDT_L], qx0[MSDT_L], qw[MSDT_L], qend[MSDT_L];
    int nq = 0;
#pragma unroll
    for (int l = 0; l < MSDT_L; ++l) {
        qy0[l] = 0; qx0[l] = 0; qw[l] = 1; qend[l] = nq;
        if (l < p.L) {
            // (32-bit unsigned: the launcher keeps every extent below 32768; 64-bit divisions here were 2 800 instructions per wave)
            const unsigned H0 = (unsigned)g.RH, W0 = (unsigned)g.RW, Hl = (unsigned)g.H[l], Wl = (unsigned)g.W[l];
            const int y0 = (int)(((unsigned)ty0 * Hl + H0 - 1u) / H0), y1 = (int)(((unsigned)ty1 * Hl + H0 - 1u) / H0);
            const int x0 = (int)(((unsigned)tx0 * Wl + W0 - 1u) / W0), x1 = (int)(((unsigned)tx1 * Wl + W0 - 1u) / W0);
            qy0[l] = y0; qx0[l] = x0; qw[l] = max(x1 - x0, 1);
            nq += max(y1 - y0, 0) * max(x1 - x0, 0);
            qend[l] = nq;
        }
    }
    if (tid < MSDT_L * 4) s_box[tid >> 2][tid & 3] = (tid & 1) ? -1 : 0x7fffffff;
    __syncthreads();
    auto query_of = [&](int idx) {                       // idx-th query of the region -> flat query index (selects: no dynamic indexing)
        int first = 0, y0 = qy0[0], x0 = qx0[0], w = qw[0], st = g.start[0], Wl = g.W[0];
#pragma unroll
        for (int k = 0; k < MSDT_L - 1; ++k)
            if (k + 1 < p.L && idx >= qend[k]) { first = qend[k]; y0 = qy0[k + 1]; x0 = qx0[k + 1]; w = qw[k + 1]; st = g.start[k + 1]; Wl = g.W[k + 1]; }
        const int r = idx - first, yy = r / w, xx = r - yy * w;
        return st + (y0 + yy) * Wl + x0 + xx;
    };
    // the four sampling locations (x, y) x 4 of level l of a query: read (forward) or derived from the projection row (backward)
    auto level_xy = [&](long row, int l, int Hl, int Wl, float4& a, float4& c) {
        if (FUSED_IN) {
            const float* off = p.offaw + row * p.ld + (long)m * LP * 2 + l * 8;
            const float4 o0 = *(const float4*)off, o1 = *(const float4*)(off + 4);
            const float2 rf = *(const float2*)(p.ref + ((p.ref_per_image ? row : row % p.Lq) * p.L + l) * 2);
            a = make_float4(rf.x + o0.x / (float)Wl, rf.y + o0.y / (float)Hl, rf.x + o0.z / (float)Wl, rf.y + o0.w / (float)Hl);
            c = make_float4(rf.x + o1.x / (float)Wl, rf.y + o1.y / (float)Hl, rf.x + o1.z / (float)Wl, rf.y + o1.w / (float)Hl);
        } else {
            const float* loc = p.loc + (row * p.M + m) * LP * 2 + l * 8;
            a = *(const float4*)loc; c = *(const float4*)(loc + 4);
        }
    };
    // pass 1: bounding boxes of the valid taps
    {
        int bx[MSDT_L][4];
#pragma unroll
        for (int l = 0; l < MSDT_L; ++l) { bx[l][0] = 0x7fffffff; bx[l][1] = -1; bx[l][2] = 0x7fffffff; bx[l][3] = -1; }
        for (int idx = tid; idx < nq; idx += nth) {
            if (g.variant & 8) break;
            const long row1 = (long)b * p.Lq + query_of(idx);
            const float* loc = FUSED_IN ? nullptr : p.loc + (row1 * p.M + m) * LP * 2;
#pragma unroll
            for (int l = 0; l < MSDT_L; ++l) {
                if (l >= p.L) continue;
                const int Hl = g.H[l], Wl = g.W[l];
                auto one = [&](float x, float y) {
                    const float him = y * Hl - 0.5f, wim = x * Wl - 0.5f;
                    if (him > -1.f && wim > -1.f && him < (float)Hl && wim < (float)Wl) {
                        const int h0 = (int)floorf(him), w0 = (int)floorf(wim);
                        bx[l][0] = min(bx[l][0], max(h0, 0)); bx[l][1] = max(bx[l][1], min(h0 + 1, Hl - 1));
                        bx[l][2] = min(bx[l][2], max(w0, 0)); bx[l][3] = max(bx[l][3], min(w0 + 1, Wl - 1));
                    }
                };
                if (P4) {
                    float4 a, c;
                    level_xy(row1, l, Hl, Wl, a, c);
                    one(a.x, a.y); one(a.z, a.w); one(c.x, c.y); one(c.z, c.w);
                } else {
                    for (int k = 0; k < p.P; ++k) { const float2 xy = *(const float2*)(loc + (l * p.P + k) * 2); one(xy.x, xy.y); }
                }
            }
        }
        // workgroup-wide min / max WITHOUT LDS atomics: all 384 threads adding to the same 12 words serialised lane by lane (190 us of a
        // 350 us launch).  Butterfly inside the wave, one row of partial results per wave, then 16 threads fold the rows.
#pragma unroll
        for (int l = 0; l < MSDT_L; ++l) {
            if (l >= p.L) continue;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                bx[l][0] = min(bx[l][0], __shfl_xor(bx[l][0], o)); bx[l][1] = max(bx[l][1], __shfl_xor(bx[l][1], o));
                bx[l][2] = min(bx[l][2], __shfl_xor(bx[l][2], o)); bx[l][3] = max(bx[l][3], __shfl_xor(bx[l][3], o));
            }
            if (lane == 0) { s_part[wave][l][0] = bx[l][0]; s_part[wave][l][1] = bx[l][1]; s_part[wave][l][2] = bx[l][2]; s_part[wave][l][3] = bx[l][3]; }
        }
    }
    __syncthreads();
    if (tid < MSDT_L * 4 && (tid >> 2) < p.L) {
        const int l = tid >> 2, k = tid & 3;
        int v = (k & 1) ? -1 : 0x7fffffff;
        for (int w = 0; w < nwave; ++w) v = (k & 1) ? max(v, s_part[w][l][k]) : min(v, s_part[w][l][k]);
        s_box[l][k] = v;
    }
    __syncthreads();
    // LDS allocation (the same decision in every thread).  Boxes that fit the budget together are staged whole; otherwise every box is
    // shrunk about its centre by one common factor (offsets that reach far leave most samples near the region all the same) and a sample
    // whose four taps are not all inside its level's box is gathered from memory.
    int by0[MSDT_L], bx0[MSDT_L], bw[MSDT_L], bh[MSDT_L], loff[MSDT_L];
    {
        long total = 0;
#pragma unroll
        for (int l = 0; l < MSDT_L; ++l) {
            by0[l] = 0; bx0[l] = 0; bw[l] = 0; bh[l] = 0; loff[l] = -1;
            if (l < p.L && __builtin_amdgcn_readfirstlane(s_box[l][1]) >= 0) {          // (readfirstlane: workgroup-uniform values, kept in SGPRs)
                by0[l] = __builtin_amdgcn_readfirstlane(s_box[l][0]); bx0[l] = __builtin_amdgcn_readfirstlane(s_box[l][2]);
                bh[l] = __builtin_amdgcn_readfirstlane(s_box[l][1]) - by0[l] + 1; bw[l] = __builtin_amdgcn_readfirstlane(s_box[l][3]) - bx0[l] + 1;
                total += (long)bh[l] * bw[l] * 64;
            }
        }
        const float f = total > g.lds_bytes ? sqrtf((float)g.lds_bytes / (float)total) * 0.98f : 1.f;
        int off = 0;
#pragma unroll
        for (int l = 0; l < MSDT_L; ++l) {
            if (bh[l] <= 0) continue;
            if (f < 1.f) {
                const int nh = max((int)((float)bh[l] * f), 1), nw = max((int)((float)bw[l] * f), 1);
                by0[l] += (bh[l] - nh) >> 1; bx0[l] += (bw[l] - nw) >> 1; bh[l] = nh; bw[l] = nw;
            }
            const int bytes = bh[l] * bw[l] * 64;
            if (off + bytes <= g.lds_bytes) { loff[l] = off; off += bytes; }
        }
    }
    if (tid < MSDT_L) {
#pragma unroll
        for (int l = 0; l < MSDT_L; ++l)
            if (tid == l) { s_lv[l][0] = by0[l]; s_lv[l][1] = bx0[l]; s_lv[l][2] = bw[l]; s_lv[l][3] = loff[l]; s_lv[l][4] = g.H[l]; s_lv[l][5] = g.W[l]; s_lv[l][6] = g.start[l]; s_lv[l][7] = bh[l]; }
    }
    const long vstride = (long)p.M * 32;
    const bf16* vbase = (const bf16*)p.value + (long)b * p.S * vstride + (long)m * 32;
    // staging by LDS-DMA (global_load_lds_dwordx4: no registers, every piece of the boxes in flight at once -- a load / ds_write loop paid
    // one L2 round trip per 6 KB).  The DMA writes lane-linearly (wave base + 16 * lane): a wave takes 16 pixels x 4 chunks of one box row,
    // and the slot swizzle is applied to the SOURCE chunk each lane fetches.
    typedef __attribute__((address_space(3))) void lds_void;
    if (!(g.variant & 1)) {
#pragma unroll
        for (int l = 0; l < MSDT_L; ++l) {
            if (loff[l] < 0) continue;
            const bf16* lv = vbase + ((long)g.start[l] + (long)by0[l] * g.W[l] + bx0[l]) * vstride;
            const int rowchunks = bw[l] * 4;
            for (int row = wave; row < bh[l]; row += nwave)
                for (int j0 = 0; j0 < rowchunks; j0 += 64) {
                    const int j = j0 + lane;
                    if (j < rowchunks) {
                        const int px = j >> 2, pix = row * bw[l] + px, c = (j & 3) ^ ((pix >> 2) & 3);
                        __builtin_amdgcn_global_load_lds(lv + ((long)row * g.W[l] + px) * vstride + c * 8,
                                                         (lds_void*)(vt + loff[l] + (row * rowchunks + j0) * 16), 16, 0, 0);
                    }
                }
        }
        __builtin_amdgcn_s_waitcnt(0x0f70);              // vmcnt(0): this wave's pieces have landed
    }
    __syncthreads();
    if (g.variant & 2) return;
    if (BWD) {
        // pass 2 (backward): d(offaw) of every (query, head) of the region
        for (int idx = tid; idx < nq; idx += nth) {
            const long row = (long)b * p.Lq + query_of(idx);
            float top[32];
            if (p.go_f32) {
                const float* gp = (const float*)p.grad_out + (row * p.M + m) * 32;
#pragma unroll
                for (int i = 0; i < 8; ++i) { const float4 t = *(const float4*)(gp + 4 * i); top[4 * i] = t.x; top[4 * i + 1] = t.y; top[4 * i + 2] = t.z; top[4 * i + 3] = t.w; }
            } else {
                const bf16* gp = (const bf16*)p.grad_out + (row * p.M + m) * 32;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const u32x4 t = *(const u32x4*)(gp + 8 * i);
#pragma unroll
                    for (int r = 0; r < 4; ++r) { top[8 * i + 2 * r] = __uint_as_float(t[r] << 16); top[8 * i + 2 * r + 1] = __uint_as_float(t[r] & 0xffff0000u); }
                }
            }
            // softmax over the L * 4 logits of this (query, head)
            const float* lg = p.offaw + row * p.ld + (long)p.M * LP * 2 + (long)m * LP;
            // (named registers, not arrays: the level loop below is not unrolled and the compiler turns selects over array elements back
            // into dynamic indexing, i.e. scratch; the exponentials are recomputed per level from the L1-resident logits instead of kept)
            float mx, inv;
            {
                const float4 lz = make_float4(-3.0e38f, -3.0e38f, -3.0e38f, -3.0e38f);
                const float4 e0 = *(const float4*)lg, e1 = p.L > 1 ? *(const float4*)(lg + 4) : lz, e2 = p.L > 2 ? *(const float4*)(lg + 8) : lz,
                             e3 = p.L > 3 ? *(const float4*)(lg + 12) : lz;
                auto max4 = [](const float4& v) { return fmaxf(fmaxf(v.x, v.y), fmaxf(v.z, v.w)); };
                mx = fmaxf(fmaxf(max4(e0), max4(e1)), fmaxf(max4(e2), max4(e3)));
                auto sum4 = [&](const float4& v) { return __expf(v.x - mx) + __expf(v.y - mx) + __expf(v.z - mx) + __expf(v.w - mx); };
                inv = 1.0f / (sum4(e0) + sum4(e1) + sum4(e2) + sum4(e3));          // (absent levels: exp(-3e38 - mx) = 0)
            }
            auto aw_of = [&](int l) {
                const float4 v = *(const float4*)(lg + 4 * l);
                return make_float4(__expf(v.x - mx) * inv, __expf(v.y - mx) * inv, __expf(v.z - mx) * inv, __expf(v.w - mx) * inv);
            };
            bf16* drow = p.doffaw + row * p.ldd;
            float4 u0 = make_float4(0.f, 0.f, 0.f, 0.f), u1 = u0, u2 = u0, u3 = u0;      // aw * d(aw) per sample
            float dot = 0.f;
#pragma unroll 1
            for (int l = 0; l < p.L; ++l) {               // (not unrolled: four inlined copies of the sample body spilled 450 bytes; selects pick the level's registers)
                const int Hl = __builtin_amdgcn_readfirstlane(s_lv[l][4]), Wl = __builtin_amdgcn_readfirstlane(s_lv[l][5]);
                const int lstart = __builtin_amdgcn_readfirstlane(s_lv[l][6]);
                const int lby0 = __builtin_amdgcn_readfirstlane(s_lv[l][0]), lbx0 = __builtin_amdgcn_readfirstlane(s_lv[l][1]);
                const int lbw = __builtin_amdgcn_readfirstlane(s_lv[l][2]), lloff = __builtin_amdgcn_readfirstlane(s_lv[l][3]);
                const int lbh = __builtin_amdgcn_readfirstlane(s_lv[l][7]);
                float4 a, c;
                level_xy(row, l, Hl, Wl, a, c);
                const float4 aw4 = aw_of(l);
                float4 ul = make_float4(0.f, 0.f, 0.f, 0.f), dW = ul, dH = ul;
#pragma unroll 1
                for (int k = 0; k < 4; ++k) {
                    const float sx = k == 0 ? a.x : k == 1 ? a.z : k == 2 ? c.x : c.z, sy = k == 0 ? a.y : k == 1 ? a.w : k == 2 ? c.y : c.w;
                    const float w = k == 0 ? aw4.x : k == 1 ? aw4.y : k == 2 ? aw4.z : aw4.w;
                    float g_w = 0.f, g_h = 0.f, g_a = 0.f;
                    const float him = sy * Hl - 0.5f, wim = sx * Wl - 0.5f;
                    if (him > -1.f && wim > -1.f && him < (float)Hl && wim < (float)Wl) {
                        const int h0 = (int)floorf(him), w0 = (int)floorf(wim);
                        const float lh = him - h0, lw = wim - w0, hh = 1.f - lh, hw = 1.f - lw;
                        const bool y0ok = h0 >= 0, y1ok = h0 + 1 <= Hl - 1, x0ok = w0 >= 0, x1ok = w0 + 1 <= Wl - 1;
                        const int ya = max(h0, 0), yb = min(h0 + 1, Hl - 1), xa = max(w0, 0), xb = min(w0 + 1, Wl - 1);
                        float d1 = 0.f, d2 = 0.f, d3 = 0.f, d4 = 0.f;
                        if (lloff >= 0 && ya >= lby0 && yb < lby0 + lbh && xa >= lbx0 && xb < lbx0 + lbw) {
                            const unsigned char* tb = vt + lloff;
                            const int i1 = (ya - lby0) * lbw + (xa - lbx0), i2 = (ya - lby0) * lbw + (xb - lbx0);
                            const int i3 = (yb - lby0) * lbw + (xa - lbx0), i4 = (yb - lby0) * lbw + (xb - lbx0);
#pragma unroll
                            for (int cc = 0; cc < 4; ++cc) {
                                msdt_dot8(*(const u32x4*)(tb + i1 * 64 + ((cc ^ ((i1 >> 2) & 3)) << 4)), top + 8 * cc, d1);
                                msdt_dot8(*(const u32x4*)(tb + i2 * 64 + ((cc ^ ((i2 >> 2) & 3)) << 4)), top + 8 * cc, d2);
                                msdt_dot8(*(const u32x4*)(tb + i3 * 64 + ((cc ^ ((i3 >> 2) & 3)) << 4)), top + 8 * cc, d3);
                                msdt_dot8(*(const u32x4*)(tb + i4 * 64 + ((cc ^ ((i4 >> 2) & 3)) << 4)), top + 8 * cc, d4);
                                if (cc & 1) __builtin_amdgcn_sched_barrier(0);      // 8 of the 16 tap reads in flight at a time (32 registers)
                            }
                        } else {
                            const bf16* lv = vbase + (long)lstart * vstride;
                            const bf16* p1 = lv + ((long)ya * Wl + xa) * vstride; const bf16* p2 = lv + ((long)ya * Wl + xb) * vstride;
                            const bf16* p3 = lv + ((long)yb * Wl + xa) * vstride; const bf16* p4 = lv + ((long)yb * Wl + xb) * vstride;
#pragma unroll
                            for (int cc = 0; cc < 4; ++cc) {
                                msdt_dot8(*(const u32x4*)(p1 + cc * 8), top + 8 * cc, d1); msdt_dot8(*(const u32x4*)(p2 + cc * 8), top + 8 * cc, d2);
                                msdt_dot8(*(const u32x4*)(p3 + cc * 8), top + 8 * cc, d3); msdt_dot8(*(const u32x4*)(p4 + cc * 8), top + 8 * cc, d4);
                                if (cc & 1) __builtin_amdgcn_sched_barrier(0);
                            }
                        }
                        // taps outside the map carry nothing (their reads were clamped to a neighbour)
                        if (!(y0ok && x0ok)) d1 = 0.f;
                        if (!(y0ok && x1ok)) d2 = 0.f;
                        if (!(y1ok && x0ok)) d3 = 0.f;
                        if (!(y1ok && x1ok)) d4 = 0.f;
                        g_a = hh * hw * d1 + hh * lw * d2 + lh * hw * d3 + lh * lw * d4;
                        g_w = (float)Wl * w * (-hh * d1 + hh * d2 - lh * d3 + lh * d4);
                        g_h = (float)Hl * w * (-hw * d1 - lw * d2 + hw * d3 + lw * d4);
                    }
                    const float ua = w * g_a, ox = g_w / (float)Wl, oy = g_h / (float)Hl;
                    if (k == 0) { ul.x = ua; dW.x = ox; dH.x = oy; } else if (k == 1) { ul.y = ua; dW.y = ox; dH.y = oy; }
                    else if (k == 2) { ul.z = ua; dW.z = ox; dH.z = oy; } else { ul.w = ua; dW.w = ox; dH.w = oy; }
                }
                if (l == 0) u0 = ul; else if (l == 1) u1 = ul; else if (l == 2) u2 = ul; else u3 = ul;
                dot += ul.x + ul.y + ul.z + ul.w;
                bf16x8 o8;
                o8[0] = (bf16)dW.x; o8[1] = (bf16)dH.x; o8[2] = (bf16)dW.y; o8[3] = (bf16)dH.y;
                o8[4] = (bf16)dW.z; o8[5] = (bf16)dH.z; o8[6] = (bf16)dW.w; o8[7] = (bf16)dH.w;
                *(bf16x8*)(drow + (long)m * LP * 2 + l * 8) = o8;
            }
            auto put = [&](int l, const float4& uu) {        // d(logit) = aw * (d(aw) - sum aw d(aw))
                const float4 ee = aw_of(l);
                bf16x4 o4;
                o4[0] = (bf16)(uu.x - ee.x * dot); o4[1] = (bf16)(uu.y - ee.y * dot);
                o4[2] = (bf16)(uu.z - ee.z * dot); o4[3] = (bf16)(uu.w - ee.w * dot);
                *(bf16x4*)(drow + (long)p.M * LP * 2 + (long)m * LP + 4 * l) = o4;
            };
            put(0, u0);
            if (p.L > 1) put(1, u1);
            if (p.L > 2) put(2, u2);
            if (p.L > 3) put(3, u3);
        }
        return;
    }
    // pass 2: the samples
    for (int idx = tid; idx < nq; idx += nth) {
        const int q = query_of(idx);
        const long grp = ((long)b * p.Lq + q) * p.M + m;
        const float* loc = FUSED_IN ? nullptr : p.loc + grp * LP * 2;
        const float* aw = FUSED_IN ? nullptr : p.attn + grp * LP;
        const long frow = (long)b * p.Lq + q;
        const float* flg = p.offaw + frow * p.ld + (long)p.M * LP * 2 + (long)m * LP;       // FUSED_IN: this (query, head)'s L * 4 logits
        float fmx = 0.f, finv = 0.f;
        if (FUSED_IN) {
            const float4 lz = make_float4(-3.0e38f, -3.0e38f, -3.0e38f, -3.0e38f);
            const float4 e0 = *(const float4*)flg, e1 = p.L > 1 ? *(const float4*)(flg + 4) : lz, e2 = p.L > 2 ? *(const float4*)(flg + 8) : lz,
                         e3 = p.L > 3 ? *(const float4*)(flg + 12) : lz;
            auto max4 = [](const float4& v) { return fmaxf(fmaxf(v.x, v.y), fmaxf(v.z, v.w)); };
            fmx = fmaxf(fmaxf(max4(e0), max4(e1)), fmaxf(max4(e2), max4(e3)));
            auto sum4 = [&](const float4& v) { return __expf(v.x - fmx) + __expf(v.y - fmx) + __expf(v.z - fmx) + __expf(v.w - fmx); };
            finv = 1.0f / (sum4(e0) + sum4(e1) + sum4(e2) + sum4(e3));
        }
        float acc[32];
#pragma unroll
        for (int i = 0; i < 32; ++i) acc[i] = 0.f;
#pragma unroll 1
        for (int l = 0; l < p.L; ++l) {                  // (not unrolled: one copy of the sample body; the level's constants come back from LDS as scalars)
            const int Hl = __builtin_amdgcn_readfirstlane(s_lv[l][4]), Wl = __builtin_amdgcn_readfirstlane(s_lv[l][5]);
            const int lstart = __builtin_amdgcn_readfirstlane(s_lv[l][6]);
            const int lby0 = __builtin_amdgcn_readfirstlane(s_lv[l][0]), lbx0 = __builtin_amdgcn_readfirstlane(s_lv[l][1]);
            const int lbw = __builtin_amdgcn_readfirstlane(s_lv[l][2]), lloff = __builtin_amdgcn_readfirstlane(s_lv[l][3]);
            const int lbh = __builtin_amdgcn_readfirstlane(s_lv[l][7]);
            float4 a, c, ww;                             // the level's four locations and weights: one round trip per level
            if (FUSED_IN) {
                level_xy(frow, l, Hl, Wl, a, c);
                const float4 v = *(const float4*)(flg + 4 * l);
                ww = make_float4(__expf(v.x - fmx) * finv, __expf(v.y - fmx) * finv, __expf(v.z - fmx) * finv, __expf(v.w - fmx) * finv);
            } else if (P4) { a = *(const float4*)(loc + l * 8); c = *(const float4*)(loc + l * 8 + 4); ww = *(const float4*)(aw + l * 4); }
            auto sample = [&](float sx, float sy, float w) {
                const float him = sy * Hl - 0.5f, wim = sx * Wl - 0.5f;
                if (!(him > -1.f && wim > -1.f && him < (float)Hl && wim < (float)Wl)) return;
                const int h0 = (int)floorf(him), w0 = (int)floorf(wim);
                const float lh = him - h0, lw = wim - w0, hh = 1.f - lh, hw = 1.f - lw;
                const bool y0ok = h0 >= 0, y1ok = h0 + 1 <= Hl - 1, x0ok = w0 >= 0, x1ok = w0 + 1 <= Wl - 1;
                // a tap outside the map keeps weight 0 and reads the (clamped) neighbouring pixel instead: its bits are multiplied by 0
                const float w1 = (y0ok && x0ok) ? hh * hw * w : 0.f, w2 = (y0ok && x1ok) ? hh * lw * w : 0.f;
                const float w3 = (y1ok && x0ok) ? lh * hw * w : 0.f, w4 = (y1ok && x1ok) ? lh * lw * w : 0.f;
                const int ya = max(h0, 0), yb = min(h0 + 1, Hl - 1), xa = max(w0, 0), xb = min(w0 + 1, Wl - 1);
                if (lloff >= 0 && ya >= lby0 && yb < lby0 + lbh && xa >= lbx0 && xb < lbx0 + lbw) {
                    const unsigned char* tb = vt + lloff;
                    const int i1 = (ya - lby0) * lbw + (xa - lbx0), i2 = (ya - lby0) * lbw + (xb - lbx0);
                    const int i3 = (yb - lby0) * lbw + (xa - lbx0), i4 = (yb - lby0) * lbw + (xb - lbx0);
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        const u32x4 v1 = *(const u32x4*)(tb + i1 * 64 + ((c ^ ((i1 >> 2) & 3)) << 4));
                        const u32x4 v2 = *(const u32x4*)(tb + i2 * 64 + ((c ^ ((i2 >> 2) & 3)) << 4));
                        const u32x4 v3 = *(const u32x4*)(tb + i3 * 64 + ((c ^ ((i3 >> 2) & 3)) << 4));
                        const u32x4 v4 = *(const u32x4*)(tb + i4 * 64 + ((c ^ ((i4 >> 2) & 3)) << 4));
                        msdt_fma8(v1, v2, v3, v4, w1, w2, w3, w4, acc + 8 * c);
                        if (c == 1) __builtin_amdgcn_sched_barrier(0);      // at most 8 of the 16 tap reads in flight (32 registers)
                    }
                } else {
                    const bf16* lv = vbase + (long)lstart * vstride;
                    const bf16* p1 = lv + ((long)ya * Wl + xa) * vstride; const bf16* p2 = lv + ((long)ya * Wl + xb) * vstride;
                    const bf16* p3 = lv + ((long)yb * Wl + xa) * vstride; const bf16* p4 = lv + ((long)yb * Wl + xb) * vstride;
#pragma unroll
                    for (int c = 0; c < 4; ++c)
                        msdt_fma8(*(const u32x4*)(p1 + c * 8), *(const u32x4*)(p2 + c * 8), *(const u32x4*)(p3 + c * 8), *(const u32x4*)(p4 + c * 8), w1, w2, w3, w4,
                                  acc + 8 * c);
                }
            };
            if (P4) {
                // one copy of the sample body per level (selects instead of four inlined copies: those kept 16 LDS reads x 4 samples live and spilled)
#pragma unroll 1
                for (int k = 0; k < 4; ++k)
                    sample(k == 0 ? a.x : k == 1 ? a.z : k == 2 ? c.x : c.z, k == 0 ? a.y : k == 1 ? a.w : k == 2 ? c.y : c.w,
                           k == 0 ? ww.x : k == 1 ? ww.y : k == 2 ? ww.z : ww.w);
            } else {
                for (int k = 0; k < p.P; ++k) {
                    const float2 xy = *(const float2*)(loc + (l * p.P + k) * 2);
                    sample(xy.x, xy.y, aw[l * p.P + k]);
                }
            }
        }
        const long o = grp * 32;
        if (p.out_f32) {
#pragma unroll
            for (int i = 0; i < 8; ++i) *(float4*)((float*)p.out + o + 4 * i) = make_float4(acc[4 * i], acc[4 * i + 1], acc[4 * i + 2], acc[4 * i + 3]);
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                bf16x8 ov;
#pragma unroll
                for (int j = 0; j < 8; ++j) ov[j] = (bf16)acc[8 * i + j];
                *(bf16x8*)((bf16*)p.out + o + 8 * i) = ov;
            }
        }
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
    int append_only;                                     // 1: d(loc) / d(attn) come from the LDS-tiled kernel; this one writes records (and adds what overflows)
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
    const bool gather = bn.append_only == 0;
    if (!gather && !any_ovf) return;

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
                if (gather) {
                    if (t.y0ok && t.x0ok) v1 = ldv4(p.value, p.v_f32, r0);
                    if (t.y0ok && t.x1ok) v2 = ldv4(p.value, p.v_f32, r0 + vstride);
                    if (t.y1ok && t.x0ok) v3 = ldv4(p.value, p.v_f32, r1);
                    if (t.y1ok && t.x1ok) v4 = ldv4(p.value, p.v_f32, r1 + vstride);
                }
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
    if (!gather) return;
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
    bn.append_only = 0;
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

// Region grid of the LDS-tiled kernels: levels from the host copy of `shapes`, regions cut from the finest map.
static int msda_tile_setup(MsdaTile& g, const int64_t* shapes_host, int B, int S, int M, int L, bool bwd, unsigned& grid, int& threads) {
    long start = 0;
    for (int l = 0; l < MSDT_L; ++l) {
        g.H[l] = 1; g.W[l] = 1; g.start[l] = 0;
        if (l >= L) continue;
        const long Hl = shapes_host[2 * l], Wl = shapes_host[2 * l + 1];
        UENC_CHECK_ARG(Hl > 0 && Wl > 0 && Hl < 32768 && Wl < 32768);
        g.H[l] = (int)Hl; g.W[l] = (int)Wl; g.start[l] = (int)start;
        start += Hl * Wl;
    }
    UENC_CHECK_ARG(start == S);                                          // level-major, nothing else in the sequence
    int ty = 8, tx = bwd ? 24 : 32, kb = 78;                             // UENC_MSDA_TILE[_BWD] = "TY,TX,KB" (tuning / tests; read per call)
    {
        const char* e = getenv(bwd ? "UENC_MSDA_TILE_BWD" : "UENC_MSDA_TILE");
        if (e && (sscanf(e, "%d,%d,%d", &ty, &tx, &kb) != 3 || ty < 1 || tx < 1 || kb < 1 || kb > 156)) { ty = 8; tx = bwd ? 24 : 32; kb = 78; }
    }
    static int lds_max = 0;
    if (kb * 1024 > lds_max) {
        const void* fns[4] = {(const void*)msda_tiled_kernel<true, 0>, (const void*)msda_tiled_kernel<false, 0>, (const void*)msda_tiled_kernel<true, 1>,
                              (const void*)msda_tiled_kernel<true, 2>};
        for (int i = 0; i < 4; ++i) {
            hipError_t er = hipFuncSetAttribute(fns[i], hipFuncAttributeMaxDynamicSharedMemorySize, kb * 1024);
            if (er != hipSuccess) return (int)er;
        }
        lds_max = kb * 1024;
    }
    g.TY = ty; g.TX = tx; g.lds_bytes = kb * 1024;
    { const char* e = getenv("UENC_MSDA_VARIANT"); g.variant = e ? atoi(e) : 0; }
    g.RH = g.H[0]; g.RW = g.W[0];
    for (int l = 1; l < L; ++l)
        if ((long)g.H[l] * g.W[l] > (long)g.RH * g.RW) { g.RH = g.H[l]; g.RW = g.W[l]; }
    const int nty = (g.RH + g.TY - 1) / g.TY;
    g.ntx = (g.RW + g.TX - 1) / g.TX;
    g.nregions = nty * g.ntx;
    // queries per region (the finest level's rectangle + the other levels' share): threads = that, rounded to waves, at most 384
    long per = 0;
    for (int l = 0; l < L; ++l) per += ((long)g.TY * g.H[l] / g.RH + 1) * ((long)g.TX * g.W[l] / g.RW + 1);
    threads = (int)((per + 63) / 64 * 64);
    const int tmax = bwd ? 256 : 384;
    threads = threads < 64 ? 64 : (threads > tmax ? tmax : threads);
    grid = (unsigned)(((long)B * g.nregions + 7) / 8 * 8 * M);
    return UENC_OK;
}

// The forward for the encoder's geometry (Lq == S: query i IS pixel i of the level-major maps), value tiles in LDS; same arguments
// and results as uenc_msdeform_attn_fwd plus the host copy of `shapes` (the grid is cut from the level-0 map).  Returns UENC_EINVAL
// -- nothing launched -- when the geometry is not the encoder's (Lq != S, D != 32, fp32 value, L > 4, L * P > 16): the caller then
// uses uenc_msdeform_attn_fwd.
extern "C" int uenc_msdeform_attn_fwd_tiled(const void* value, int v_dtype, const int64_t* shapes, const int64_t* level_start,
                                            const float* loc, const float* attn, void* out, int out_dtype, int B, int S, int M,
                                            int D, int L, int Lq, int P, const int64_t* shapes_host, hipStream_t stream) {
    MsdaP p;
    int rc = msda_fill(p, value, v_dtype, shapes, level_start, loc, attn, B, S, M, D, L, Lq, P);
    if (rc != UENC_OK) return rc;
    UENC_CHECK_ARG(out && shapes_host && D == 32 && v_dtype == UENC_BF16 && Lq == S && L <= MSDT_L && L * P <= 16 && (long)B * M < 65536 && (long)B * S * M < (1L << 31) / 64);
    UENC_CHECK_ARG(((uintptr_t)out & 15) == 0 && ((uintptr_t)loc & 7) == 0);
    p.out = out; p.out_f32 = (out_dtype == UENC_F32);
    MsdaTile g;
    unsigned grid = 0; int threads = 0;
    rc = msda_tile_setup(g, shapes_host, B, S, M, L, false, grid, threads);
    if (rc != UENC_OK) return rc;
    if (P == 4 && ((uintptr_t)loc & 15) == 0 && ((uintptr_t)attn & 15) == 0)
        hipLaunchKernelGGL((msda_tiled_kernel<true, 0>), dim3(grid), dim3(threads), g.lds_bytes, stream, p, g);
    else
        hipLaunchKernelGGL((msda_tiled_kernel<false, 0>), dim3(grid), dim3(threads), g.lds_bytes, stream, p, g);
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

// The fused forward for the encoder's geometry, value tiles in LDS (see uenc_msdeform_attn_fwd_tiled): locations and softmaxed weights are
// derived inside the kernel, so neither uenc_msda_prep_fwd nor its two tensors exist.  Needs P == 4, 16-byte aligned offaw rows (ld % 4 == 0);
// returns -1 (nothing launched) when the geometry is not eligible -- call uenc_msdeform_attn_fused_fwd then.
extern "C" int uenc_msdeform_attn_fused_fwd_tiled(const void* value, int v_dtype, const int64_t* shapes, const int64_t* level_start, const float* offaw,
                                                  long ld, const float* ref, int ref_per_image, void* out, int out_dtype, int B, int S, int M, int D,
                                                  int L, int Lq, int P, const int64_t* shapes_host, hipStream_t stream) {
    MsdaP p;
    int rc = msda_fill_fused(p, value, v_dtype, shapes, level_start, offaw, ld, ref, ref_per_image, B, S, M, D, L, Lq, P);
    if (rc != UENC_OK) return rc;
    UENC_CHECK_ARG(out && shapes_host && D == 32 && v_dtype == UENC_BF16 && Lq == S && L <= MSDT_L && P == 4 && (long)B * M < 65536 &&
                   (long)B * S * M < (1L << 31) / 64);
    UENC_CHECK_ARG(((uintptr_t)out & 15) == 0 && ((uintptr_t)offaw & 15) == 0 && ld % 4 == 0 && ((uintptr_t)ref & 7) == 0);
    p.out = out; p.out_f32 = (out_dtype == UENC_F32);
    MsdaTile g;
    unsigned grid = 0; int threads = 0;
    rc = msda_tile_setup(g, shapes_host, B, S, M, L, false, grid, threads);
    if (rc != UENC_OK) return rc;
    hipLaunchKernelGGL((msda_tiled_kernel<true, 2>), dim3(grid), dim3(threads), g.lds_bytes, stream, p, g);
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
    // encoder geometry (queries = the maps' pixels, P == 4, 16-byte aligned rows): d(offaw) from the LDS-tiled kernel, the binned kernel only
    // appends its records.  UENC_MSDA_TILED_BWD=0: the one-kernel form (A/B).
    bool tiled = Lq == S && P == 4 && L <= MSDT_L && v_dtype == UENC_BF16 && (long)B * M < 65536 && (long)B * S * M < (1L << 31) / 64 &&
                 ((uintptr_t)offaw & 15) == 0 && ld % 4 == 0 && ((uintptr_t)doffaw & 15) == 0 && ld_doffaw % 8 == 0 && ((uintptr_t)ref & 7) == 0;
    { const char* ev = getenv("UENC_MSDA_TILED_BWD"); if (ev && atoi(ev) == 0) tiled = false; }
    MsdaTile g;
    unsigned tgrid = 0; int tthreads = 0;
    if (tiled && msda_tile_setup(g, shapes_host, B, S, M, L, true, tgrid, tthreads) != UENC_OK) tiled = false;
    bn.append_only = tiled ? 1 : 0;
    hipLaunchKernelGGL(msda_bwd_bin_kernel<true>, dim3((unsigned)(nchunk * B * M)), dim3(256), 0, stream, p, bn);
    if (tiled) hipLaunchKernelGGL((msda_tiled_kernel<true, 1>), dim3(tgrid), dim3(tthreads), g.lds_bytes, stream, p, g);
    if (p.go_f32) hipLaunchKernelGGL(msda_bin_reduce_kernel<true>, dim3((unsigned)((nitems + 3) / 4)), dim3(256), 0, stream, p, bn);
    else hipLaunchKernelGGL(msda_bin_reduce_kernel<false>, dim3((unsigned)((nitems + 3) / 4)), dim3(256), 0, stream, p, bn);
    UENC_LAUNCH_RET();
}
