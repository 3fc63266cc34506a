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
// Backward: one lane per channel so each grad_value atomic wave-instruction covers two 128-byte
// row segments (the full-rate atomic shape, MI355X_MICROARCH.md "Global float atomics");
// grad_loc / grad_attn are reduced over channels with wave shuffles (no LDS, no atomics).
// `value` may be fp32 (the reference's contract) or bf16 (half the gather bytes).
#include "common.h"

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
};

__device__ __forceinline__ float4 ldv4(const void* base, int is_f32, long idx) {
    if (is_f32) return *(const float4*)((const float*)base + idx);
    const bf16x4 v = *(const bf16x4*)((const bf16*)base + idx);
    return make_float4((float)v[0], (float)v[1], (float)v[2], (float)v[3]);
}

template <int LPG>
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
    const float* loc = p.loc + grp * LP * 2;
    const float* aw = p.attn + grp * LP;
    const long vstride = (long)p.M * p.D;               // between spatial positions
    const long vbase = (long)b * p.S * vstride + (long)m * p.D + c4;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int l = 0; l < p.L; ++l) {
        const int Hl = (int)p.shapes[2 * l], Wl = (int)p.shapes[2 * l + 1];
        const long lbase = vbase + p.level_start[l] * vstride;
        for (int k = 0; k < p.P; ++k) {
            const float x = loc[(l * p.P + k) * 2], y = loc[(l * p.P + k) * 2 + 1];
            const float w = aw[l * p.P + k];
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
    }
    const long o = grp * p.D + c4;
    if (p.out_f32) *(float4*)((float*)p.out + o) = acc;
    else {
        bf16x4 ov; ov[0] = (bf16)acc.x; ov[1] = (bf16)acc.y; ov[2] = (bf16)acc.z; ov[3] = (bf16)acc.w;
        *(bf16x4*)((bf16*)p.out + o) = ov;
    }
}

// backward: D lanes per (b,q,m) (D = 32 or 64), lane = channel
template <int D>
__global__ __launch_bounds__(256) void msda_bwd_kernel(MsdaP p) {
    const long gtid = (long)blockIdx.x * 256 + threadIdx.x;
    const long grp = gtid / D;
    const int c = (int)(gtid - grp * D);
    const long ngrp = (long)p.B * p.Lq * p.M;
    if (grp >= ngrp) return;       // D divides 64: whole groups leave together
    const int m = (int)(grp % p.M);
    const long bq = grp / p.M;
    const int b = (int)(bq / p.Lq);
    const int LP = p.L * p.P;
    const float* loc = p.loc + grp * LP * 2;
    const float* aw = p.attn + grp * LP;
    const long vstride = (long)p.M * D;
    const long vbase = (long)b * p.S * vstride + (long)m * D + c;
    const float top = p.go_f32 ? ((const float*)p.grad_out)[grp * D + c] : (float)((const bf16*)p.grad_out)[grp * D + c];
    auto ldv = [&](long idx) -> float {
        return p.v_f32 ? ((const float*)p.value)[idx] : (float)((const bf16*)p.value)[idx];
    };
    for (int l = 0; l < p.L; ++l) {
        const int Hl = (int)p.shapes[2 * l], Wl = (int)p.shapes[2 * l + 1];
        const long lbase = vbase + p.level_start[l] * vstride;
        for (int k = 0; k < p.P; ++k) {
            const float x = loc[(l * p.P + k) * 2], y = loc[(l * p.P + k) * 2 + 1];
            const float w = aw[l * p.P + k];
            const float him = y * Hl - 0.5f, wim = x * Wl - 0.5f;
            float g_w = 0.f, g_h = 0.f, g_a = 0.f;
            if (him > -1.f && wim > -1.f && him < (float)Hl && wim < (float)Wl) {
                const int h0 = (int)floorf(him), w0 = (int)floorf(wim);
                const float lh = him - h0, lw = wim - w0, hh = 1.f - lh, hw = 1.f - lw;
                const float tg = top * w;
                const long r0 = lbase + ((long)h0 * Wl + w0) * vstride;
                float v1 = 0.f, v2 = 0.f, v3 = 0.f, v4 = 0.f;
                if (h0 >= 0 && w0 >= 0) { v1 = ldv(r0); atomicAdd(p.grad_value + r0, hh * hw * tg); }
                if (h0 >= 0 && w0 + 1 <= Wl - 1) { v2 = ldv(r0 + vstride); atomicAdd(p.grad_value + r0 + vstride, hh * lw * tg); }
                if (h0 + 1 <= Hl - 1 && w0 >= 0) {
                    v3 = ldv(r0 + (long)Wl * vstride);
                    atomicAdd(p.grad_value + r0 + (long)Wl * vstride, lh * hw * tg);
                }
                if (h0 + 1 <= Hl - 1 && w0 + 1 <= Wl - 1) {
                    v4 = ldv(r0 + (long)Wl * vstride + vstride);
                    atomicAdd(p.grad_value + r0 + (long)Wl * vstride + vstride, lh * lw * tg);
                }
                g_a = top * (hh * hw * v1 + hh * lw * v2 + lh * hw * v3 + lh * lw * v4);
                g_w = (float)Wl * tg * (-hh * v1 + hh * v2 - lh * v3 + lh * v4);
                g_h = (float)Hl * tg * (-hw * v1 - lw * v2 + hw * v3 + lw * v4);
            }
#pragma unroll
            for (int o = D / 2; o > 0; o >>= 1) {
                g_w += __shfl_xor(g_w, o);
                g_h += __shfl_xor(g_h, o);
                g_a += __shfl_xor(g_a, o);
            }
            if (c == 0) {
                p.grad_loc[(grp * LP + l * p.P + k) * 2] = g_w;
                p.grad_loc[(grp * LP + l * p.P + k) * 2 + 1] = g_h;
                p.grad_attn[grp * LP + l * p.P + k] = g_a;
            }
        }
    }
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
    if (D == 32) hipLaunchKernelGGL(msda_fwd_kernel<8>, dim3(grid), dim3(256), 0, stream, p);
    else if (D == 16) hipLaunchKernelGGL(msda_fwd_kernel<4>, dim3(grid), dim3(256), 0, stream, p);
    else hipLaunchKernelGGL(msda_fwd_kernel<16>, dim3(grid), dim3(256), 0, stream, p);
    UENC_LAUNCH_RET();
}

// Mirrors ms_deform_attn_backward(...): grad_value must be zero-filled by the caller (it is accumulated);
// grad_loc / grad_attn are fully overwritten.
extern "C" int uenc_msdeform_attn_bwd(const void* value, int v_dtype, const int64_t* shapes, const int64_t* level_start,
                                      const float* loc, const float* attn, const void* grad_out, int go_dtype,
                                      float* grad_value, float* grad_loc, float* grad_attn, int B, int S, int M, int D, int L,
                                      int Lq, int P, hipStream_t stream) {
    MsdaP p;
    int rc = msda_fill(p, value, v_dtype, shapes, level_start, loc, attn, B, S, M, D, L, Lq, P);
    if (rc != UENC_OK) return rc;
    UENC_CHECK_ARG(grad_out && grad_value && grad_loc && grad_attn && (D == 32 || D == 64 || D == 16));
    p.grad_out = grad_out; p.go_f32 = (go_dtype == UENC_F32);
    p.grad_value = grad_value; p.grad_loc = grad_loc; p.grad_attn = grad_attn;
    const long threads = (long)B * Lq * M * D;
    const unsigned grid = (unsigned)((threads + 255) / 256);
    if (D == 32) hipLaunchKernelGGL(msda_bwd_kernel<32>, dim3(grid), dim3(256), 0, stream, p);
    else if (D == 64) hipLaunchKernelGGL(msda_bwd_kernel<64>, dim3(grid), dim3(256), 0, stream, p);
    else hipLaunchKernelGGL(msda_bwd_kernel<16>, dim3(grid), dim3(256), 0, stream, p);
    UENC_LAUNCH_RET();
}
