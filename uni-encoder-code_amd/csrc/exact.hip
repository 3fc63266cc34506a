// fp32 "exact" arithmetic mode of the hot path ("HIP fp32 mode <= 1e-4").
//
// The product's kernels feed the matrix cores bf16 operands.  This file is the same path with every operand, every
// activation handed from kernel to kernel and every accumulation in fp32, selected by UENC_EXACT=1 (uenc/ops.py): the mode
// that shows what part of a deviation from the fp32 reference is bf16 rounding and what part would be a bug.  It is a
// verification mode, built for correctness first: the GEMM runs on the fp32-input matrix instruction
// v_mfma_f32_16x16x4_f32 (exact fp32 products and sums, 1/16 of the bf16 rate), the two attention cores are plain VALU
// kernels with per-lane online softmax.  Same C-ABI conventions as the rest of the library (include/uenc.h).
//
//   uenc_gemm_nt_f32          replaces the same ATen calls as uenc_gemm_nt (Linear / 1x1 conv / mask einsum, forward + dgrad)
//   uenc_gemm_tn_f32          weight / bias gradient dW += dY^T X, db += colsum(dY)
//   uenc_window_attn_f32_fwd  model/modeling/backbone/swin.py:250-289 around WindowAttention.forward :131-171
//   uenc_window_attn_f32_bwd  its backward
//   uenc_mha_f32_fwd / _bwd   the nn.MultiheadAttention cores of transformer_decoder/*.py (head_dim 32)
#include "common.h"
#include <math.h>

enum { XEPI_NONE = 0, XEPI_GELU = 1, XEPI_RELU = 2, XEPI_RESIDUAL = 3, XEPI_MUL_DGELU = 4, XEPI_MUL_DRELU = 5 };

__device__ __forceinline__ float gelu_exact(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752f)); }
__device__ __forceinline__ float dgelu_exact(float x) {
    return 0.5f * (1.0f + erff(x * 0.70710678118654752f)) + x * expf(-0.5f * x * x) * 0.39894228040143268f;
}

// ---------------------------------------------------------------------------------------------------------------------
// C[m][n] = epi(alpha * (sum_k A[m][k] * W[n][k] + bias[n])), everything fp32.  TRANS_A: A is stored [K][M] and W [K][N]
// (the token-contraction form dW[n][k] = sum_m dY[m][n] X[m][k] with A := dY, W := X read column-wise), used by gemm_tn_f32.
// 128 x 128 tile, BK 16, four waves of 64 x 64; operands k-major in LDS ([16][128 + 16]: fragment reads of one MFMA hit 32
// different banks).
// ---------------------------------------------------------------------------------------------------------------------
struct XGemm {
    const float* A; const float* W; float* C; const float* bias; const float* aux; float* aux_out;
    long lda, ldw, ldc, ldaux, ldaux_out;
    int M, N, K, epi, accumulate;
    float alpha;
    float* colsum;          // TRANS only: colsum[m] += sum_k A[k][m] (bias gradient), added by the n-tile-0 workgroups
};

constexpr int XBM = 128, XBN = 128, XBK = 16, XLD = XBM + 16;

template <bool TRANS>
__global__ __launch_bounds__(256) void gemm_f32_kernel(XGemm p) {
    __shared__ float As[XBK][XLD];
    __shared__ float Ws[XBK][XLD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int m0 = blockIdx.y * XBM, n0 = blockIdx.x * XBN;
    const int wm = (wave >> 1) * 64, wn = (wave & 1) * 64;
    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float csum = 0.f;       // TRANS: column sum of A for column m0 + (tid & 127), half of the k range per thread half

    float4 ra[2], rw[2];
    auto load = [&](int k0) {
        if (!TRANS) {
            // row = tid & 127, two 4-wide k groups per thread: 32 consecutive rows per LDS store instruction
            const int row = tid & 127, kq0 = (tid >> 7) * 2;
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int k = k0 + (kq0 + q) * 4;
                ra[q] = (m0 + row < p.M && k < p.K) ? *(const float4*)(p.A + (long)(m0 + row) * p.lda + k) : make_float4(0.f, 0.f, 0.f, 0.f);
                rw[q] = (n0 + row < p.N && k < p.K) ? *(const float4*)(p.W + (long)(n0 + row) * p.ldw + k) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        } else {
            // A[k][m]: k = tid >> 5 (+8), 4 consecutive m per thread
            const int c4 = (tid & 31) * 4;
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int k = k0 + (tid >> 5) + q * 8;
                ra[q] = (k < p.K && m0 + c4 < p.M) ? *(const float4*)(p.A + (long)k * p.lda + m0 + c4) : make_float4(0.f, 0.f, 0.f, 0.f);
                rw[q] = (k < p.K && n0 + c4 < p.N) ? *(const float4*)(p.W + (long)k * p.ldw + n0 + c4) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
    };
    auto store = [&]() {
        if (!TRANS) {
            const int row = tid & 127, kq0 = (tid >> 7) * 2;
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int kk = (kq0 + q) * 4;
                As[kk + 0][row] = ra[q].x; As[kk + 1][row] = ra[q].y; As[kk + 2][row] = ra[q].z; As[kk + 3][row] = ra[q].w;
                Ws[kk + 0][row] = rw[q].x; Ws[kk + 1][row] = rw[q].y; Ws[kk + 2][row] = rw[q].z; Ws[kk + 3][row] = rw[q].w;
            }
        } else {
            const int c4 = (tid & 31) * 4;
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int kk = (tid >> 5) + q * 8;
                *(float4*)&As[kk][c4] = ra[q];
                *(float4*)&Ws[kk][c4] = rw[q];
            }
        }
    };

    load(0);
    for (int k0 = 0; k0 < p.K; k0 += XBK) {
        __syncthreads();
        store();
        __syncthreads();
        if (k0 + XBK < p.K) load(k0 + XBK);
        if (TRANS && p.colsum != nullptr && blockIdx.x == 0) {
            const int c = tid & 127, kh = (tid >> 7) * 8;
#pragma unroll
            for (int kk = 0; kk < 8; ++kk) csum += As[kh + kk][c];
        }
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const int kk = ks * 4 + (lane >> 4), r = lane & 15;
            float a[4], b[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) { a[i] = As[kk][wm + i * 16 + r]; b[i] = Ws[kk][wn + i * 16 + r]; }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], b[j], acc[i][j], 0, 0, 0);
        }
    }
    if (TRANS && p.colsum != nullptr && blockIdx.x == 0) {
        __syncthreads();
        float* red = &As[0][0];
        red[tid] = csum;
        __syncthreads();
        if (tid < 128 && m0 + tid < p.M) p.colsum[m0 + tid] += red[tid] + red[tid + 128];
    }
    // epilogue: lane l, register r of tile (i, j) = C[wm + 16 i + 4 (l >> 4) + r][wn + 16 j + (l & 15)]
    const int col = lane & 15, rq = (lane >> 4) * 4;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int n = n0 + wn + j * 16 + col;
            if (n >= p.N) continue;
            const float bv = p.bias != nullptr ? p.bias[n] : 0.f;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = m0 + wm + i * 16 + rq + r;
                if (m >= p.M) continue;
                float v = (acc[i][j][r] + bv) * p.alpha;
                if (p.epi == XEPI_GELU) {
                    if (p.aux_out != nullptr) p.aux_out[(long)m * p.ldaux_out + n] = v;
                    v = gelu_exact(v);
                } else if (p.epi == XEPI_RELU) {
                    v = fmaxf(v, 0.f);
                } else if (p.epi == XEPI_RESIDUAL) {
                    v += p.aux[(long)m * p.ldaux + n];
                } else if (p.epi == XEPI_MUL_DGELU) {
                    v *= dgelu_exact(p.aux[(long)m * p.ldaux + n]);
                } else if (p.epi == XEPI_MUL_DRELU) {
                    v = p.aux[(long)m * p.ldaux + n] > 0.f ? v : 0.f;
                }
                float* c = p.C + (long)m * p.ldc + n;
                *c = p.accumulate ? *c + v : v;
            }
        }
}

extern "C" int uenc_gemm_nt_f32(const float* A, long lda, const float* W, long ldw, float* C, long ldc, int M, int N, int K,
                                const float* bias, int epilogue, const float* aux, long ldaux, float* aux_out, long ldaux_out,
                                float alpha, int accumulate, hipStream_t stream) {
    UENC_CHECK_ARG(A != nullptr && W != nullptr && C != nullptr && M > 0 && N > 0 && K > 0);
    UENC_CHECK_ARG(K % 4 == 0 && lda % 4 == 0 && ldw % 4 == 0 && ((uintptr_t)A % 16) == 0 && ((uintptr_t)W % 16) == 0);
    UENC_CHECK_ARG(epilogue >= XEPI_NONE && epilogue <= XEPI_MUL_DRELU);
    if (epilogue >= XEPI_RESIDUAL) UENC_CHECK_ARG(aux != nullptr);
    if (accumulate) UENC_CHECK_ARG(epilogue == XEPI_NONE);
    XGemm p{A, W, C, bias, aux, aux_out, lda, ldw, ldc, ldaux, ldaux_out, M, N, K, epilogue, accumulate, alpha, nullptr};
    dim3 grid((N + XBN - 1) / XBN, (M + XBM - 1) / XBM);
    hipLaunchKernelGGL(gemm_f32_kernel<false>, grid, dim3(256), 0, stream, p);
    UENC_LAUNCH_RET();
}

// dW[n][k] += sum_m dY[m][n] * X[m][k]; db[n] += sum_m dY[m][n]   (dY (M, N), X (M, K) row-major fp32; dW (N, K), db fp32)
extern "C" int uenc_gemm_tn_f32(const float* dY, long ldy, const float* X, long ldx, float* dW, long ldw, float* db, int M, int N, int K,
                                hipStream_t stream) {
    UENC_CHECK_ARG(dY != nullptr && X != nullptr && dW != nullptr && M > 0 && N > 0 && K > 0);
    UENC_CHECK_ARG(N % 4 == 0 && K % 4 == 0 && ldy % 4 == 0 && ldx % 4 == 0 && ((uintptr_t)dY % 16) == 0 && ((uintptr_t)X % 16) == 0);
    // output rows = N (columns of dY), output columns = K (columns of X), contraction over the M tokens
    XGemm p{dY, X, dW, nullptr, nullptr, nullptr, ldy, ldx, ldw, 0, 0, N, K, M, XEPI_NONE, 1, 1.0f, db};
    dim3 grid((K + XBN - 1) / XBN, (N + XBM - 1) / XBM);
    hipLaunchKernelGGL(gemm_f32_kernel<true>, grid, dim3(256), 0, stream, p);
    UENC_LAUNCH_RET();
}

// ---------------------------------------------------------------------------------------------------------------------
// window attention, fp32.  One workgroup per (window, head); thread t owns query row t of the window.
// ---------------------------------------------------------------------------------------------------------------------
struct XWin {
    const float* qkv; const float* qkv_bias; const float* table; float* out;
    const float* dout; float* dqkv; float* dtable; float* dbias_pad;      // backward
    int B, H, W, C, nH, ws, shift, Hp, Wp, nWw, nWin, N;
    float scale;
};

__device__ __forceinline__ int xregion3(int v, int P, int ws, int shift) { return (v >= P - ws) + (v >= P - shift); }

// tokoff: flat token index of window slot t, -1 for a padding slot (zero row after norm1: its q / k / v are the qkv bias)
__device__ __forceinline__ void xwin_slots(const XWin& p, int b, int wi, int wj, int* tokoff, int* rid) {
    for (int t = threadIdx.x; t < p.N; t += blockDim.x) {
        const int ty = t / p.ws, tx = t - ty * p.ws;
        const int hs = wi * p.ws + ty, wx = wj * p.ws + tx;
        int ho = hs + p.shift, wo = wx + p.shift;
        if (ho >= p.Hp) ho -= p.Hp;
        if (wo >= p.Wp) wo -= p.Wp;
        tokoff[t] = (ho < p.H && wo < p.W) ? (b * p.H + ho) * p.W + wo : -1;
        rid[t] = p.shift > 0 ? 3 * xregion3(hs, p.Hp, p.ws, p.shift) + xregion3(wx, p.Wp, p.ws, p.shift) : 0;
    }
}

constexpr int XHD = 32, XHP = 33;       // head_dim, padded LDS row

__device__ __forceinline__ void xwin_stage(const XWin& p, int head, const int* tokoff, float* Qs, float* Ks, float* Vs) {
    for (int idx = threadIdx.x; idx < p.N * XHD; idx += blockDim.x) {
        const int t = idx >> 5, d = idx & 31;
        const int off = tokoff[t];
        const float* src = off >= 0 ? p.qkv + (long)off * 3 * p.C : p.qkv_bias;
        Qs[t * XHP + d] = src[head * XHD + d];
        Ks[t * XHP + d] = src[p.C + head * XHD + d];
        Vs[t * XHP + d] = src[2 * p.C + head * XHD + d];
    }
}

__device__ __forceinline__ float xwin_bias(const XWin& p, int head, int t, int j, const int* rid) {
    const int ty = t / p.ws, tx = t - ty * p.ws, jy = j / p.ws, jx = j - jy * p.ws;
    const int idx = (ty - jy + p.ws - 1) * (2 * p.ws - 1) + (tx - jx + p.ws - 1);
    return p.table[(long)idx * p.nH + head] + (rid[t] != rid[j] ? -100.0f : 0.0f);
}

__global__ void xwin_fwd_kernel(XWin p) {
    extern __shared__ float xs[];
    float* Qs = xs; float* Ks = Qs + p.N * XHP; float* Vs = Ks + p.N * XHP;
    int* tokoff = (int*)(Vs + p.N * XHP); int* rid = tokoff + p.N;
    const int head = blockIdx.x % p.nH, win = blockIdx.x / p.nH;
    const int b = win / p.nWin, wr = win - b * p.nWin, wi = wr / p.nWw, wj = wr - wi * p.nWw;
    xwin_slots(p, b, wi, wj, tokoff, rid);
    __syncthreads();
    xwin_stage(p, head, tokoff, Qs, Ks, Vs);
    __syncthreads();
    const int t = threadIdx.x;
    if (t >= p.N || tokoff[t] < 0) return;
    float q[XHD], acc[XHD];
#pragma unroll
    for (int d = 0; d < XHD; ++d) { q[d] = Qs[t * XHP + d] * p.scale; acc[d] = 0.f; }
    float m = -INFINITY, l = 0.f;
    for (int j = 0; j < p.N; ++j) {
        float s = 0.f;
#pragma unroll
        for (int d = 0; d < XHD; ++d) s = fmaf(q[d], Ks[j * XHP + d], s);
        s += xwin_bias(p, head, t, j, rid);
        const float mn = fmaxf(m, s), c = expf(m - mn), e = expf(s - mn);
        l = l * c + e;
#pragma unroll
        for (int d = 0; d < XHD; ++d) acc[d] = fmaf(e, Vs[j * XHP + d], acc[d] * c);
        m = mn;
    }
    const float inv = 1.0f / l;
    float* o = p.out + (long)tokoff[t] * p.C + head * XHD;
#pragma unroll
    for (int d = 0; d < XHD; ++d) o[d] = acc[d] * inv;
}

// backward: thread t recomputes row t's probabilities (two passes: max / sum, then gradients), writes dq directly and adds its
// dk / dv / d(table) contributions with LDS / global float atomics (a verification kernel: order-dependent at fp32 rounding level)
__global__ void xwin_bwd_kernel(XWin p) {
    extern __shared__ float xs[];
    float* Qs = xs; float* Ks = Qs + p.N * XHP; float* Vs = Ks + p.N * XHP;
    float* dKs = Vs + p.N * XHP; float* dVs = dKs + p.N * XHP;
    int* tokoff = (int*)(dVs + p.N * XHP); int* rid = tokoff + p.N;
    const int head = blockIdx.x % p.nH, win = blockIdx.x / p.nH;
    const int b = win / p.nWin, wr = win - b * p.nWin, wi = wr / p.nWw, wj = wr - wi * p.nWw;
    xwin_slots(p, b, wi, wj, tokoff, rid);
    for (int idx = threadIdx.x; idx < p.N * XHP; idx += blockDim.x) { dKs[idx] = 0.f; dVs[idx] = 0.f; }
    __syncthreads();
    xwin_stage(p, head, tokoff, Qs, Ks, Vs);
    __syncthreads();
    const int t = threadIdx.x;
    // padding-slot queries produce outputs that are cropped away: their dO is zero, they contribute nothing
    if (t < p.N && tokoff[t] >= 0) {
        float q[XHD], go[XHD], dq[XHD];
        const float* gp = p.dout + (long)tokoff[t] * p.C + head * XHD;
#pragma unroll
        for (int d = 0; d < XHD; ++d) { q[d] = Qs[t * XHP + d] * p.scale; go[d] = gp[d]; dq[d] = 0.f; }
        float m = -INFINITY;
        for (int j = 0; j < p.N; ++j) {
            float s = 0.f;
#pragma unroll
            for (int d = 0; d < XHD; ++d) s = fmaf(q[d], Ks[j * XHP + d], s);
            m = fmaxf(m, s + xwin_bias(p, head, t, j, rid));
        }
        float l = 0.f, delta = 0.f;         // delta = sum_j P_j * (dO . v_j)
        for (int j = 0; j < p.N; ++j) {
            float s = 0.f, dp = 0.f;
#pragma unroll
            for (int d = 0; d < XHD; ++d) { s = fmaf(q[d], Ks[j * XHP + d], s); dp = fmaf(go[d], Vs[j * XHP + d], dp); }
            const float e = expf(s + xwin_bias(p, head, t, j, rid) - m);
            l += e;
            delta = fmaf(e, dp, delta);
        }
        const float inv = 1.0f / l;
        delta *= inv;
        const int ty = t / p.ws, tx = t - ty * p.ws;
        for (int jj = 0; jj < p.N; ++jj) {
            int j = jj + t;                                  // staggered start: the lanes of a wave add to different dK / dV rows
            if (j >= p.N) j -= p.N;
            float s = 0.f, dp = 0.f;
#pragma unroll
            for (int d = 0; d < XHD; ++d) { s = fmaf(q[d], Ks[j * XHP + d], s); dp = fmaf(go[d], Vs[j * XHP + d], dp); }
            const float pr = expf(s + xwin_bias(p, head, t, j, rid) - m) * inv;
            const float ds = pr * (dp - delta);             // d(score), also the table gradient of this (t, j) pair
#pragma unroll
            for (int d = 0; d < XHD; ++d) {
                dq[d] = fmaf(ds, Ks[j * XHP + d], dq[d]);
                atomicAdd(&dKs[j * XHP + d], ds * q[d]);
                atomicAdd(&dVs[j * XHP + d], pr * go[d]);
            }
            if (p.dtable != nullptr) {
                const int jy = j / p.ws, jx = j - jy * p.ws;
                atomicAdd(p.dtable + (long)((ty - jy + p.ws - 1) * (2 * p.ws - 1) + (tx - jx + p.ws - 1)) * p.nH + head, ds);
            }
        }
        float* o = p.dqkv + (long)tokoff[t] * 3 * p.C + head * XHD;
#pragma unroll
        for (int d = 0; d < XHD; ++d) o[d] = dq[d] * p.scale;
    }
    __syncthreads();
    for (int idx = threadIdx.x; idx < p.N * XHD; idx += blockDim.x) {
        const int j = idx >> 5, d = idx & 31;
        const int off = tokoff[j];
        if (off >= 0) {
            float* o = p.dqkv + (long)off * 3 * p.C + head * XHD + d;
            o[p.C] = dKs[j * XHP + d];
            o[2 * p.C] = dVs[j * XHP + d];
        } else if (p.dbias_pad != nullptr) {        // a padding slot's k / v are the bias itself
            atomicAdd(p.dbias_pad + p.C + head * XHD + d, dKs[j * XHP + d]);
            atomicAdd(p.dbias_pad + 2 * p.C + head * XHD + d, dVs[j * XHP + d]);
        }
    }
}

static int xwin_fill(XWin& p, int B, int H, int W, int C, int nH, int ws, int shift, float scale) {
    UENC_CHECK_ARG(B > 0 && H > 0 && W > 0 && C == nH * XHD && ws >= 2 && ws <= 16 && shift >= 0 && shift < ws);
    p.B = B; p.H = H; p.W = W; p.C = C; p.nH = nH; p.ws = ws; p.shift = shift; p.scale = scale;
    p.Hp = (H + ws - 1) / ws * ws; p.Wp = (W + ws - 1) / ws * ws;
    p.nWw = p.Wp / ws; p.nWin = (p.Hp / ws) * p.nWw; p.N = ws * ws;
    return UENC_OK;
}

extern "C" int uenc_window_attn_f32_fwd(const float* qkv, const float* qkv_bias, const float* table, float* out, int B, int H, int W, int C,
                                        int nH, int ws, int shift, float scale, hipStream_t stream) {
    UENC_CHECK_ARG(qkv != nullptr && qkv_bias != nullptr && table != nullptr && out != nullptr);
    XWin p{qkv, qkv_bias, table, out, nullptr, nullptr, nullptr, nullptr};
    if (int e = xwin_fill(p, B, H, W, C, nH, ws, shift, scale)) return e;
    const int nth = (p.N + 63) / 64 * 64;
    const size_t smem = (size_t)(3 * p.N * XHP + 2 * p.N) * 4;
    hipLaunchKernelGGL(xwin_fwd_kernel, dim3(B * p.nWin * nH), dim3(nth), smem, stream, p);
    UENC_LAUNCH_RET();
}

// dqkv (B, H, W, 3C) is fully written; dtable ((2ws-1)^2, nH) and dbias_pad (3C) are ACCUMULATED into (atomics).
extern "C" int uenc_window_attn_f32_bwd(const float* qkv, const float* qkv_bias, const float* table, const float* dout, float* dqkv,
                                        float* dtable, float* dbias_pad, int B, int H, int W, int C, int nH, int ws, int shift, float scale,
                                        hipStream_t stream) {
    UENC_CHECK_ARG(qkv != nullptr && qkv_bias != nullptr && table != nullptr && dout != nullptr && dqkv != nullptr);
    XWin p{qkv, qkv_bias, table, nullptr, dout, dqkv, dtable, dbias_pad};
    if (int e = xwin_fill(p, B, H, W, C, nH, ws, shift, scale)) return e;
    const int nth = (p.N + 63) / 64 * 64;
    const size_t smem = (size_t)(5 * p.N * XHP + 2 * p.N) * 4;
    UENC_CHECK_ARG(smem <= 160 * 1024);
    if (smem > 64 * 1024) (void)hipFuncSetAttribute((const void*)xwin_bwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    hipLaunchKernelGGL(xwin_bwd_kernel, dim3(B * p.nWin * nH), dim3(nth), smem, stream, p);
    UENC_LAUNCH_RET();
}

// ---------------------------------------------------------------------------------------------------------------------
// multi-head attention core, fp32, head_dim 32: q (B, Lq, E), k / v (B, S, E) with element strides (batch, row), heads
// interleaved in E; mask (B, Lq, mrs) bytes, 1 = blocked.  One workgroup = 8 queries of one (image, head); each of its four
// waves owns two of them; key tiles of 64 are staged in LDS and every LANE keeps its own running softmax over the keys it
// sees (lane l handles keys l, l + 64, ...): no cross-lane traffic until the final merge.
// ---------------------------------------------------------------------------------------------------------------------
struct XMha {
    const float* q; const float* k; const float* v; const unsigned char* mask; float* out; float* lse;
    const float* dout; float* dq; float* dk; float* dv; const float* delta;     // backward
    long qs0, qs1, ks0, ks1, vs0, vs1, os0, os1, mrs;
    long gos0, gos1, dqs0, dqs1, dks0, dks1, dvs0, dvs1;
    int B, nH, Lq, S;
    float scale;
    unsigned drop_thresh, seed;
    float inv_keep;
};

constexpr int XQW = 2, XQB = 8;

__global__ __launch_bounds__(256) void xmha_fwd_kernel(XMha p) {
    __shared__ float Ks[64 * XHP];
    __shared__ float Vs[64 * XHP];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.z, h = blockIdx.y, q0 = blockIdx.x * XQB + wave * XQW;
    float qv[XQW][XHD], acc[XQW][XHD], m[XQW], l[XQW];
#pragma unroll
    for (int i = 0; i < XQW; ++i) {
        const int qi = min(q0 + i, p.Lq - 1);
        const float* qp = p.q + b * p.qs0 + qi * p.qs1 + h * XHD;
#pragma unroll
        for (int d = 0; d < XHD; ++d) { qv[i][d] = qp[d] * p.scale; acc[i][d] = 0.f; }
        m[i] = -INFINITY; l[i] = 0.f;
    }
    for (int s0 = 0; s0 < p.S; s0 += 64) {
        __syncthreads();
        for (int idx = tid; idx < 64 * XHD; idx += 256) {
            const int r = idx >> 5, d = idx & 31;
            const bool ok = s0 + r < p.S;
            Ks[r * XHP + d] = ok ? p.k[b * p.ks0 + (long)(s0 + r) * p.ks1 + h * XHD + d] : 0.f;
            Vs[r * XHP + d] = ok ? p.v[b * p.vs0 + (long)(s0 + r) * p.vs1 + h * XHD + d] : 0.f;
        }
        __syncthreads();
        const int key = s0 + lane;
        if (key >= p.S) continue;
        float kr[XHD], vr[XHD];
#pragma unroll
        for (int d = 0; d < XHD; ++d) { kr[d] = Ks[lane * XHP + d]; vr[d] = Vs[lane * XHP + d]; }
#pragma unroll
        for (int i = 0; i < XQW; ++i) {
            const int qi = q0 + i;
            if (qi >= p.Lq) continue;
            if (p.mask != nullptr && p.mask[((long)b * p.Lq + qi) * p.mrs + key]) continue;
            float s = 0.f;
#pragma unroll
            for (int d = 0; d < XHD; ++d) s = fmaf(qv[i][d], kr[d], s);
            const float mn = fmaxf(m[i], s), c = expf(m[i] - mn), e = expf(s - mn);
            l[i] = l[i] * c + e;
            float ed = e;                                   // dropout after the normaliser
            if (p.drop_thresh != 0u)
                ed = attn_keep(p.seed, p.drop_thresh, (((unsigned long long)b * p.nH + h) * p.Lq + qi) * p.S + key) ? e * p.inv_keep : 0.f;
#pragma unroll
            for (int d = 0; d < XHD; ++d) acc[i][d] = fmaf(ed, vr[d], acc[i][d] * c);
            m[i] = mn;
        }
    }
#pragma unroll
    for (int i = 0; i < XQW; ++i) {
        const int qi = q0 + i;
        if (qi >= p.Lq) continue;                         // wave-uniform
        const float M = wave_max(m[i]);
        const float w = m[i] == -INFINITY ? 0.f : expf(m[i] - M);
        const float L = wave_sum(l[i] * w);
        float* o = p.out + b * p.os0 + qi * p.os1 + h * XHD;
#pragma unroll
        for (int d = 0; d < XHD; ++d) {
            const float v = wave_sum(acc[i][d] * w);
            if (lane == 0) o[d] = v / L;
        }
        if (lane == 0 && p.lse != nullptr) p.lse[((long)b * p.nH + h) * p.Lq + qi] = M + logf(L);
    }
}

// backward: same tiling; probabilities from the saved log-sum-exp; dq summed over lanes at the end, dk / dv added with atomics
// (the (b, h) slices of dk / dv are shared by the query blocks).  delta[b][h][q] = dO . O is computed by the caller's
// elementwise pass (uenc_mha_f32_delta).
__global__ __launch_bounds__(256) void xmha_bwd_kernel(XMha p) {
    __shared__ float Ks[64 * XHP];
    __shared__ float Vs[64 * XHP];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.z, h = blockIdx.y, q0 = blockIdx.x * XQB + wave * XQW;
    float qv[XQW][XHD], go[XQW][XHD], dq[XQW][XHD], lse[XQW], dl[XQW];
#pragma unroll
    for (int i = 0; i < XQW; ++i) {
        const int qi = min(q0 + i, p.Lq - 1);
        const float* qp = p.q + b * p.qs0 + qi * p.qs1 + h * XHD;
        const float* gp = p.dout + b * p.gos0 + qi * p.gos1 + h * XHD;
#pragma unroll
        for (int d = 0; d < XHD; ++d) { qv[i][d] = qp[d] * p.scale; go[i][d] = gp[d]; dq[i][d] = 0.f; }
        lse[i] = p.lse[((long)b * p.nH + h) * p.Lq + qi];
        dl[i] = p.delta[((long)b * p.nH + h) * p.Lq + qi];
    }
    for (int s0 = 0; s0 < p.S; s0 += 64) {
        __syncthreads();
        for (int idx = tid; idx < 64 * XHD; idx += 256) {
            const int r = idx >> 5, d = idx & 31;
            const bool ok = s0 + r < p.S;
            Ks[r * XHP + d] = ok ? p.k[b * p.ks0 + (long)(s0 + r) * p.ks1 + h * XHD + d] : 0.f;
            Vs[r * XHP + d] = ok ? p.v[b * p.vs0 + (long)(s0 + r) * p.vs1 + h * XHD + d] : 0.f;
        }
        __syncthreads();
        const int key = s0 + lane;
        if (key >= p.S) continue;
        float kr[XHD], vr[XHD], dkr[XHD], dvr[XHD];
#pragma unroll
        for (int d = 0; d < XHD; ++d) { kr[d] = Ks[lane * XHP + d]; vr[d] = Vs[lane * XHP + d]; dkr[d] = 0.f; dvr[d] = 0.f; }
        bool any = false;
#pragma unroll
        for (int i = 0; i < XQW; ++i) {
            const int qi = q0 + i;
            if (qi >= p.Lq) continue;
            if (p.mask != nullptr && p.mask[((long)b * p.Lq + qi) * p.mrs + key]) continue;
            float s = 0.f, dp = 0.f;
#pragma unroll
            for (int d = 0; d < XHD; ++d) { s = fmaf(qv[i][d], kr[d], s); dp = fmaf(go[i][d], vr[d], dp); }
            const float pr = expf(s - lse[i]);
            float keepw = 1.0f;
            if (p.drop_thresh != 0u)
                keepw = attn_keep(p.seed, p.drop_thresh, (((unsigned long long)b * p.nH + h) * p.Lq + qi) * p.S + key) ? p.inv_keep : 0.f;
            const float ds = pr * (dp * keepw - dl[i]);
#pragma unroll
            for (int d = 0; d < XHD; ++d) {
                dq[i][d] = fmaf(ds, kr[d], dq[i][d]);
                dkr[d] = fmaf(ds, qv[i][d], dkr[d]);
                dvr[d] = fmaf(pr * keepw, go[i][d], dvr[d]);
            }
            any = true;
        }
        if (any) {
            float* dkp = p.dk + b * p.dks0 + (long)key * p.dks1 + h * XHD;
            float* dvp = p.dv + b * p.dvs0 + (long)key * p.dvs1 + h * XHD;
#pragma unroll
            for (int d = 0; d < XHD; ++d) { atomicAdd(dkp + d, dkr[d]); atomicAdd(dvp + d, dvr[d]); }
        }
    }
#pragma unroll
    for (int i = 0; i < XQW; ++i) {
        const int qi = q0 + i;
        if (qi >= p.Lq) continue;
        float* o = p.dq + b * p.dqs0 + qi * p.dqs1 + h * XHD;
#pragma unroll
        for (int d = 0; d < XHD; ++d) {
            const float v = wave_sum(dq[i][d]);
            if (lane == 0) o[d] = v * p.scale;
        }
    }
}

__global__ void xmha_delta_kernel(const float* out, long os0, long os1, const float* dout, long gs0, long gs1, float* delta, int B, int nH, int Lq) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)B * nH * Lq) return;
    const int q = (int)(i % Lq), h = (int)((i / Lq) % nH), b = (int)(i / ((long)Lq * nH));
    const float* o = out + b * os0 + q * os1 + h * XHD;
    const float* g = dout + b * gs0 + q * gs1 + h * XHD;
    float s = 0.f;
#pragma unroll
    for (int d = 0; d < XHD; ++d) s = fmaf(o[d], g[d], s);
    delta[i] = s;
}

extern "C" int uenc_mha_f32_fwd(const float* q, long qs0, long qs1, const float* k, long ks0, long ks1, const float* v, long vs0, long vs1,
                                const uint8_t* mask, long mask_row_stride, float* out, long os0, long os1, float* lse, int B, int nH, int Lq,
                                int S, float scale, float dropout_p, unsigned seed, hipStream_t stream) {
    UENC_CHECK_ARG(q != nullptr && k != nullptr && v != nullptr && out != nullptr && B > 0 && nH > 0 && Lq > 0 && S > 0);
    UENC_CHECK_ARG(dropout_p >= 0.f && dropout_p < 1.f);
    if (mask != nullptr) UENC_CHECK_ARG(mask_row_stride >= S);
    XMha p{};
    p.q = q; p.k = k; p.v = v; p.mask = mask; p.out = out; p.lse = lse;
    p.qs0 = qs0; p.qs1 = qs1; p.ks0 = ks0; p.ks1 = ks1; p.vs0 = vs0; p.vs1 = vs1; p.os0 = os0; p.os1 = os1; p.mrs = mask_row_stride;
    p.B = B; p.nH = nH; p.Lq = Lq; p.S = S; p.scale = scale;
    p.drop_thresh = attn_drop_thresh(dropout_p); p.seed = seed; p.inv_keep = 1.0f / (1.0f - dropout_p);
    hipLaunchKernelGGL(xmha_fwd_kernel, dim3((Lq + XQB - 1) / XQB, nH, B), dim3(256), 0, stream, p);
    UENC_LAUNCH_RET();
}

// dq is fully written; dk / dv (B, S, E) must be ZEROED by the caller (atomics).  delta: (B, nH, Lq) fp32 scratch.
extern "C" int uenc_mha_f32_bwd(const float* q, long qs0, long qs1, const float* k, long ks0, long ks1, const float* v, long vs0, long vs1,
                                const uint8_t* mask, long mask_row_stride, const float* out, long os0, long os1, const float* lse,
                                const float* dout, long gos0, long gos1, float* dq, long dqs0, long dqs1, float* dk, long dks0, long dks1,
                                float* dv, long dvs0, long dvs1, float* delta, int B, int nH, int Lq, int S, float scale, float dropout_p,
                                unsigned seed, hipStream_t stream) {
    UENC_CHECK_ARG(dropout_p >= 0.f && dropout_p < 1.f);
    UENC_CHECK_ARG(q != nullptr && k != nullptr && v != nullptr && out != nullptr && lse != nullptr && dout != nullptr);
    UENC_CHECK_ARG(dq != nullptr && dk != nullptr && dv != nullptr && delta != nullptr && B > 0 && nH > 0 && Lq > 0 && S > 0);
    if (mask != nullptr) UENC_CHECK_ARG(mask_row_stride >= S);
    const long n = (long)B * nH * Lq;
    hipLaunchKernelGGL(xmha_delta_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, out, os0, os1, dout, gos0, gos1, delta, B, nH, Lq);
    XMha p{};
    p.q = q; p.k = k; p.v = v; p.mask = mask; p.lse = const_cast<float*>(lse); p.dout = dout; p.dq = dq; p.dk = dk; p.dv = dv; p.delta = delta;
    p.qs0 = qs0; p.qs1 = qs1; p.ks0 = ks0; p.ks1 = ks1; p.vs0 = vs0; p.vs1 = vs1; p.mrs = mask_row_stride;
    p.gos0 = gos0; p.gos1 = gos1; p.dqs0 = dqs0; p.dqs1 = dqs1; p.dks0 = dks0; p.dks1 = dks1; p.dvs0 = dvs0; p.dvs1 = dvs1;
    p.B = B; p.nH = nH; p.Lq = Lq; p.S = S; p.scale = scale;
    p.drop_thresh = attn_drop_thresh(dropout_p); p.seed = seed; p.inv_keep = 1.0f / (1.0f - dropout_p);
    hipLaunchKernelGGL(xmha_bwd_kernel, dim3((Lq + XQB - 1) / XQB, nH, B), dim3(256), 0, stream, p);
    UENC_LAUNCH_RET();
}
