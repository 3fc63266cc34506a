// bf16 MFMA GEMMs with fused prologues / epilogues (gfx950).
//
//   uenc_gemm_nt :  C[m][n] = epi( alpha * (sum_k A[m][k] * W[n][k] + bias[n]) )      (forward, dgrad)
//   uenc_gemm_tn :  dW[n][k] += sum_m dY[m][n] * X[m][k] ;  db[n] += sum_m dY[m][n]    (wgrad)
//
// These carry every Linear of the path: qkv / proj / fc1 / fc2 of the Swin blocks
// (reference model/modeling/backbone/swin.py:35-41, 138-170), PatchMerging.reduction (:335),
// the deformable encoder's projections and FFN (pixel_decoder/ops/modules/ms_deform_attn.py:103-125,
// pixel_decoder/msdeformattn.py:126-130), nn.MultiheadAttention in/out projections, decoder FFNs and
// the mask einsum (transformer_decoder/oneformer_transformer_decoder.py:183, 498-500).
//
// Tile 128x128x64, 4 waves (2x2), each wave 64x64 = 4x4 MFMA 16x16x32 tiles, fp32 accumulators.
// The MFMA A operand is the W fragment and the B operand the X fragment, so a lane's 4 accumulator
// registers are 4 consecutive n of one output row: 8-byte (bf16) / 16-byte (fp32) stores.
// Global -> registers -> LDS staging with the loads of tile t+1 issued before the MFMAs of tile t
// and written after them (one barrier per k-step, two LDS buffers).  LDS rows are 128 B (64 bf16);
// 16-byte chunk c of row r lives at chunk c ^ ((r >> 1) & 7): conflict-free ds_read_b128 fragments
// and conflict-free ds_write_b128 staging.
#include "common.h"
#include "prof.h"
#include <stdlib.h>

#define BM 128
#define BN 128
#define BK 64
#define GEMM_THREADS 256

enum { EPI_NONE = 0, EPI_GELU = 1, EPI_RELU = 2, EPI_RESIDUAL = 3, EPI_MUL_DGELU = 4, EPI_MUL_DRELU = 5 };

struct GemmNT {
    const void* A; int a_f32; long lda;
    const bf16* W; long ldw;
    void* C; long ldc;
    const float* bias;
    const void* aux; long ldaux;
    bf16* aux_out; long ldaux_out;
    int M, N, K;
    int tiles_m, tiles_n;
    int klen;      // K elements handled by one split (multiple of BK); == K rounded up when splitk == 1
    int atomic;    // fp32 atomicAdd into C (split-K or accumulate)
    int variant;   // debug A/B switch
    int persist;   // gemm_nt256_kernel: the grid is smaller than the tile count, workgroups walk tile positions
    float alpha;
    const float* sample_scale; float inv_rows;   // optional per-sample multiplier of alpha: row m belongs to sample floor(m / rows_per_sample)
    long part_stride;     // > 0: split-K with STORED partials: split y writes its tile to C + y * part_stride (fp32 elements), no atomics
    int splits;           // gridDim.y
    long bsA, bsW, bsC;   // batched form (gemm_nt_kernel only): element strides between the problems of blockIdx.z
    // LayerNorm fused into the fp32 residual epilogue of the LDS-staged kernels when ONE column tile covers the row (N <= 256): the 32
    // threads that share a row of the staged tile reduce it, and y = LN(C row) leaves as fp32 and / or bf16 beside C (the pre-norm sum the
    // LayerNorm backward needs) and the row's (mean, rstd).  ln_gamma == nullptr: off.
    const float* ln_gamma; const float* ln_beta; float* ln_y32; bf16* ln_y16; float2* ln_stats; float ln_eps;
};

// alpha of output row m: p.alpha, times the row's sample scale when one is given (DropPath: 0 or 1 / keep_prob per image)
__device__ __forceinline__ float nt_alpha(const GemmNT& p, int m) {
    return p.sample_scale == nullptr ? p.alpha : p.alpha * p.sample_scale[(int)(((float)m + 0.5f) * p.inv_rows)];
}

__device__ __forceinline__ int lds_off(int row, int chunk) { return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4); }

// Load 8 consecutive k of one row as bf16x8, zero outside [0, nrows) x [0, kend).  Branch-free: the address is
// clamped into the matrix and the result masked, so a thread's loads issue back to back.
template <bool F32>
__device__ __forceinline__ u32x4 load_row8(const void* base, long ld, int row, int nrows, int k, int kend, int K) {
    const bool ok = (row < nrows) && (k < kend);
    const int rc = min(row, nrows - 1), kc = min(k, K - 8);
    u32x4 v;
    if (F32) {
        const float* p = (const float*)base + (long)rc * ld + kc;
        const float4 a = *(const float4*)p, b = *(const float4*)(p + 4);
        const bf16x8 r = cvt8(a, b);
        v = *(const u32x4*)&r;
    } else {
        v = *(const u32x4*)((const bf16*)base + (long)rc * ld + kc);
    }
    const u32x4 z = {0u, 0u, 0u, 0u};
    return ok ? v : z;
}

template <int EPI, int OUT_F32>
__global__ __launch_bounds__(GEMM_THREADS) void gemm_nt_kernel(GemmNT p) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * (BM + BN) * BK * 2];
    if (blockIdx.z > 0) {                        // batched launch: same shapes, operands / output at fixed strides
        p.A = (const char*)p.A + (long)blockIdx.z * p.bsA * (p.a_f32 ? 4 : 2);
        p.W += (long)blockIdx.z * p.bsW;
        p.C = (char*)p.C + (long)blockIdx.z * p.bsC * (OUT_F32 ? 4 : 2);
    }
    if (OUT_F32 && p.part_stride > 0) p.C = (float*)p.C + (long)blockIdx.y * p.part_stride;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int tile = xcd_remap(blockIdx.x, gridDim.x);
    const int mt = tile / p.tiles_n, nt = tile - mt * p.tiles_n;
    const int m0 = mt * BM, n0 = nt * BN;
    const int kbeg = blockIdx.y * p.klen;
    const int kend = min(p.K, kbeg + p.klen);
    const int nkt = (kend - kbeg + BK - 1) / BK;

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int srow = t >> 3, schunk = t & 7;
    u32x4 ra[4], rw[4];
    auto gload = [&](int kt) {
        const int k = kbeg + kt * BK + schunk * 8;
#pragma unroll
        for (int i = 0; i < 4; ++i) rw[i] = load_row8<false>(p.W, p.ldw, n0 + srow + 32 * i, p.N, k, kend, p.K);
        if (p.a_f32) {
#pragma unroll
            for (int i = 0; i < 4; ++i) ra[i] = load_row8<true>(p.A, p.lda, m0 + srow + 32 * i, p.M, k, kend, p.K);
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) ra[i] = load_row8<false>(p.A, p.lda, m0 + srow + 32 * i, p.M, k, kend, p.K);
        }
    };
    auto lstore = [&](int buf) {
        unsigned char* As = smem + buf * ((BM + BN) * BK * 2);
        unsigned char* Ws = As + BM * BK * 2;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            *(u32x4*)(As + lds_off(srow + 32 * i, schunk)) = ra[i];
            *(u32x4*)(Ws + lds_off(srow + 32 * i, schunk)) = rw[i];
        }
    };

    gload(0);
    lstore(0);
    __syncthreads();
    const int fr = lane & 15, fg = lane >> 4;
    for (int kt = 0; kt < nkt; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < nkt) gload(kt + 1);
        const unsigned char* As = smem + buf * ((BM + BN) * BK * 2);
        const unsigned char* Ws = As + BM * BK * 2;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 wf[4], xf[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) wf[i] = *(const bf16x8*)(Ws + lds_off(wn * 64 + i * 16 + fr, ks * 4 + fg));
#pragma unroll
            for (int j = 0; j < 4; ++j) xf[j] = *(const bf16x8*)(As + lds_off(wm * 64 + j * 16 + fr, ks * 4 + fg));
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = mfma16(wf[i], xf[j], acc[i][j]);
        }
        if (kt + 1 < nkt) lstore(buf ^ 1);
        __syncthreads();
    }

    // ---- epilogue ----
    // split-K / accumulate: fp32 atomics straight from the accumulators (lane holds C[m = ..+fr][n = ..+4fg+0..3])
    const bool lead = (blockIdx.y == 0);
    if (p.atomic) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int m = m0 + wm * 64 + j * 16 + fr;
            if (m >= p.M) continue;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int n = n0 + wn * 64 + i * 16 + 4 * fg;
                if (n >= p.N) continue;
                float* c = (float*)p.C + (long)m * p.ldc + n;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float v = acc[i][j][r];
                    if (p.bias != nullptr && lead) v += p.bias[n + r];
                    atomicAdd(c + r, v * p.alpha);
                }
            }
        }
        return;
    }
    // Coalesced path: the fp32 tile goes through LDS (two 64-row halves, rows padded to 132 floats so the
    // accumulator writes are bank-conflict free), then every thread owns 8 consecutive columns of a row: bias /
    // residual / saved activations are read and the result written as full 128-byte-per-16-lanes row segments.
    float* T = (float*)smem;
    constexpr int LDT = 132;
    const int erow = t >> 4, ecol = (t & 15) * 8;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        __syncthreads();
        if (wm == h) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int i = 0; i < 4; ++i) *(f32x4*)(T + (j * 16 + fr) * LDT + wn * 64 + i * 16 + 4 * fg) = acc[i][j];
        }
        __syncthreads();
        const int n = n0 + ecol;
        if (n >= p.N) continue;
        const bool full8 = (n + 8 <= p.N);                 // N % 4 == 0: either 8 or 4 valid columns
        float bv[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if (p.bias != nullptr) {
            const float4 b0 = *(const float4*)(p.bias + n);
            bv[0] = b0.x; bv[1] = b0.y; bv[2] = b0.z; bv[3] = b0.w;
            if (full8) { const float4 b1 = *(const float4*)(p.bias + n + 4); bv[4] = b1.x; bv[5] = b1.y; bv[6] = b1.z; bv[7] = b1.w; }
        }
#pragma unroll
        for (int ps = 0; ps < 4; ++ps) {
            const int rl = ps * 16 + erow;
            const int m = m0 + h * 64 + rl;
            if (m >= p.M) continue;
            const f32x4 a0 = *(const f32x4*)(T + rl * LDT + ecol), a1 = *(const f32x4*)(T + rl * LDT + ecol + 4);
            float v[8];
            const float al = nt_alpha(p, m);
#pragma unroll
            for (int r = 0; r < 4; ++r) { v[r] = (a0[r] + bv[r]) * al; v[4 + r] = (a1[r] + bv[4 + r]) * al; }
            if (EPI == EPI_GELU) {
                if (p.aux_out != nullptr) {
                    bf16x8 pre;
#pragma unroll
                    for (int r = 0; r < 8; ++r) pre[r] = (bf16)v[r];
                    bf16* dst = p.aux_out + (long)m * p.ldaux_out + n;
                    if (full8) *(bf16x8*)dst = pre;
                    else { bf16x4 q4; q4[0] = pre[0]; q4[1] = pre[1]; q4[2] = pre[2]; q4[3] = pre[3]; *(bf16x4*)dst = q4; }
                }
#pragma unroll
                for (int r = 0; r < 8; ++r) v[r] = gelu_f(v[r]);
            } else if (EPI == EPI_RELU) {
#pragma unroll
                for (int r = 0; r < 8; ++r) v[r] = fmaxf(v[r], 0.f);
            } else if (EPI == EPI_RESIDUAL) {
                const float* rp = (const float*)p.aux + (long)m * p.ldaux + n;
                const float4 r0 = *(const float4*)rp;
                v[0] += r0.x; v[1] += r0.y; v[2] += r0.z; v[3] += r0.w;
                if (full8) { const float4 r1 = *(const float4*)(rp + 4); v[4] += r1.x; v[5] += r1.y; v[6] += r1.z; v[7] += r1.w; }
            } else if (EPI == EPI_MUL_DGELU || EPI == EPI_MUL_DRELU) {
                const bf16* ap = (const bf16*)p.aux + (long)m * p.ldaux + n;
                bf16x8 sv;
                if (full8) sv = *(const bf16x8*)ap;
                else {
                    const bf16x4 q4 = *(const bf16x4*)ap;
#pragma unroll
                    for (int r = 0; r < 4; ++r) { sv[r] = q4[r]; sv[4 + r] = (bf16)0.f; }
                }
#pragma unroll
                for (int r = 0; r < 8; ++r)
                    v[r] = (EPI == EPI_MUL_DGELU) ? v[r] * dgelu_f((float)sv[r]) : (((float)sv[r] > 0.f) ? v[r] : 0.f);
            }
            if (OUT_F32) {
                float* c = (float*)p.C + (long)m * p.ldc + n;
                *(float4*)c = make_float4(v[0], v[1], v[2], v[3]);
                if (full8) *(float4*)(c + 4) = make_float4(v[4], v[5], v[6], v[7]);
            } else {
                bf16x8 o;
#pragma unroll
                for (int r = 0; r < 8; ++r) o[r] = (bf16)v[r];
                bf16* c = (bf16*)p.C + (long)m * p.ldc + n;
                if (full8) *(bf16x8*)c = o;
                else { bf16x4 q4; q4[0] = o[0]; q4[1] = o[1]; q4[2] = o[2]; q4[3] = o[3]; *(bf16x4*)c = q4; }
            }
        }
    }
}


// ---------------------------------------------------------------------------------------------
// Skinny variant for the GEMMs with a handful of output tiles (the transformer decoder: M = B * 150 query rows, N = 256 ... 2048,
// K up to 2048; ~250 launches per step that ran 6 ... 48 workgroups of the 128-tile kernel for 10 ... 25 us each, every k-tile
// paying a full memory latency).  Here a workgroup owns a 64 x 64 tile and its four waves split the K range between them:
// fragments go global -> registers directly in MFMA layout (nothing is shared between waves, so no LDS staging), 4 k-steps
// (32 loads per lane) in flight per iteration, the four partial tiles are summed through LDS and the epilogue runs once.
// ---------------------------------------------------------------------------------------------
template <int EPI, int OUT_F32>
__device__ __forceinline__ void epilogue8(const GemmNT& p, int m, int n, bool full8, float (&v)[8]) {
    if (EPI == EPI_GELU) {
        if (p.aux_out != nullptr) {
            bf16x8 pre;
#pragma unroll
            for (int r = 0; r < 8; ++r) pre[r] = (bf16)v[r];
            bf16* dst = p.aux_out + (long)m * p.ldaux_out + n;
            if (full8) *(bf16x8*)dst = pre;
            else { bf16x4 q4; q4[0] = pre[0]; q4[1] = pre[1]; q4[2] = pre[2]; q4[3] = pre[3]; *(bf16x4*)dst = q4; }
        }
#pragma unroll
        for (int r = 0; r < 8; ++r) v[r] = gelu_f(v[r]);
    } else if (EPI == EPI_RELU) {
#pragma unroll
        for (int r = 0; r < 8; ++r) v[r] = fmaxf(v[r], 0.f);
    } else if (EPI == EPI_RESIDUAL) {
        const float* rp = (const float*)p.aux + (long)m * p.ldaux + n;
        const float4 r0 = *(const float4*)rp;
        v[0] += r0.x; v[1] += r0.y; v[2] += r0.z; v[3] += r0.w;
        if (full8) { const float4 r1 = *(const float4*)(rp + 4); v[4] += r1.x; v[5] += r1.y; v[6] += r1.z; v[7] += r1.w; }
    } else if (EPI == EPI_MUL_DGELU || EPI == EPI_MUL_DRELU) {
        const bf16* ap = (const bf16*)p.aux + (long)m * p.ldaux + n;
        bf16x8 sv;
        if (full8) sv = *(const bf16x8*)ap;
        else {
            const bf16x4 q4 = *(const bf16x4*)ap;
#pragma unroll
            for (int r = 0; r < 4; ++r) { sv[r] = q4[r]; sv[4 + r] = (bf16)0.f; }
        }
#pragma unroll
        for (int r = 0; r < 8; ++r)
            v[r] = (EPI == EPI_MUL_DGELU) ? v[r] * dgelu_f((float)sv[r]) : (((float)sv[r] > 0.f) ? v[r] : 0.f);
    }
    if (OUT_F32) {
        float* c = (float*)p.C + (long)m * p.ldc + n;
        *(float4*)c = make_float4(v[0], v[1], v[2], v[3]);
        if (full8) *(float4*)(c + 4) = make_float4(v[4], v[5], v[6], v[7]);
    } else {
        bf16x8 o;
#pragma unroll
        for (int r = 0; r < 8; ++r) o[r] = (bf16)v[r];
        bf16* c = (bf16*)p.C + (long)m * p.ldc + n;
        if (full8) *(bf16x8*)c = o;
        else { bf16x4 q4; q4[0] = o[0]; q4[1] = o[1]; q4[2] = o[2]; q4[3] = o[3]; *(bf16x4*)c = q4; }
    }
}

template <int EPI, int OUT_F32, int A_F32>
__global__ __launch_bounds__(256) void gemm_nt_skinny_kernel(GemmNT p) {
    constexpr int LDP = 68;                                   // 64 + 4: the accumulator writes are bank-conflict free
    __shared__ __attribute__((aligned(16))) float part[4 * 64 * LDP];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int fr = lane & 15, fg = lane >> 4;
    const int mt = blockIdx.x / p.tiles_n, nt = blockIdx.x - mt * p.tiles_n;
    const int m0 = mt * 64, n0 = nt * 64;
    const int per = (((p.K + 31) >> 5) + 3) >> 2;             // 32-wide k-steps per wave
    const int kb = wave * per * 32, ke = min(p.K, kb + per * 32);

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    long aoff[4], woff[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        aoff[j] = (long)min(m0 + j * 16 + fr, p.M - 1) * p.lda;
        woff[j] = (long)min(n0 + j * 16 + fr, p.N - 1) * p.ldw;
    }
    const u32x4 z = {0u, 0u, 0u, 0u};
    for (int k = kb; k < ke; k += 128) {
        u32x4 xa[4][4], wa[4][4];
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int kk = k + s * 32 + 8 * fg;
            const bool ok = kk < ke;
            const int kc = min(kk, p.K - 8);                  // clamped address, masked value: the loads stay unconditional
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (A_F32) {
                    const float* ap = (const float*)p.A + aoff[j] + kc;
                    const float4 a0 = *(const float4*)ap, a1 = *(const float4*)(ap + 4);
                    const bf16x8 r = cvt8(a0, a1);
                    xa[s][j] = ok ? *(const u32x4*)&r : z;
                } else {
                    const u32x4 v = *(const u32x4*)((const bf16*)p.A + aoff[j] + kc);
                    xa[s][j] = ok ? v : z;
                }
                const u32x4 w = *(const u32x4*)(p.W + woff[j] + kc);
                wa[s][j] = ok ? w : z;
            }
        }
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = mfma16(*(const bf16x8*)&wa[s][i], *(const bf16x8*)&xa[s][j], acc[i][j]);
    }
    // acc[i][j][r] = C[m = j*16 + fr][n = i*16 + 4*fg + r]
    float* mine = part + wave * 64 * LDP;
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int i = 0; i < 4; ++i) *(f32x4*)(mine + (j * 16 + fr) * LDP + i * 16 + 4 * fg) = acc[i][j];
    __syncthreads();
    const int row = t >> 2, m = m0 + row;
    if (m >= p.M) return;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int c = (t & 3) * 16 + h * 8, n = n0 + c;
        if (n >= p.N) continue;
        const bool full8 = (n + 8 <= p.N);
        float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            const f32x4 a0 = *(const f32x4*)(part + (w * 64 + row) * LDP + c), a1 = *(const f32x4*)(part + (w * 64 + row) * LDP + c + 4);
#pragma unroll
            for (int r = 0; r < 4; ++r) { v[r] += a0[r]; v[4 + r] += a1[r]; }
        }
        if (p.bias != nullptr) {
            const float4 b0 = *(const float4*)(p.bias + n);
            v[0] += b0.x; v[1] += b0.y; v[2] += b0.z; v[3] += b0.w;
            if (full8) { const float4 b1 = *(const float4*)(p.bias + n + 4); v[4] += b1.x; v[5] += b1.y; v[6] += b1.z; v[7] += b1.w; }
        }
#pragma unroll
        for (int r = 0; r < 8; ++r) v[r] *= nt_alpha(p, m);
        epilogue8<EPI, OUT_F32>(p, m, n, full8, v);
    }
}

// ---------------------------------------------------------------------------------------------
// Large-tile variant for the GEMMs that carry most of the FLOPs (stage 3 / 4 of the backbone,
// K % 64 == 0, bf16 A): 256 x 256 x 64 tile, 8 waves (2 x 4), each wave 128(m) x 64(n) = 8 x 4
// MFMA tiles (128 accumulator VGPRs), one workgroup per CU.  Operands go global -> LDS directly
// (global_load_lds_dwordx4: no staging registers).  LDS-DMA writes lane-linearly (wave base +
// lane * 16 B), so the XOR swizzle that keeps the ds_read_b128 fragment reads conflict-free is
// applied to the per-lane SOURCE address (cdna_hip_programming.md rule 21).
// Main loop PIPE = 1 (default): the staggered quadrant pipeline described in the kernel (half-tile
// DMA stream six half-tiles ahead, counted vmcnt, raw s_barrier, the two waves of a SIMD one
// barrier apart): 1.30 PFLOP/s at 8192^3 on random operands, +3...14 % over PIPE = 0 on the
// workload's shapes.  PIPE = 0: two 64 KB stages, the DMA of k-tile t+1 issued before the 64
// MFMAs per wave of k-tile t and drained by the __syncthreads() after them (kept for A/B:
// UENC_GEMM_VARIANT bit 128).
// ---------------------------------------------------------------------------------------------
#define BM2 256
#define BN2 256
#define T2 512

// counted wait on the VM counter (LDS-DMA and global loads / stores in flight): at most n (0 .. 8) of this wave's operations stay in flight
__device__ __forceinline__ void wait_vmcnt_upto8(int n) {
    if (n >= 8) __builtin_amdgcn_s_waitcnt(0x0f78);
    else if (n == 7) __builtin_amdgcn_s_waitcnt(0x0f77);
    else if (n == 6) __builtin_amdgcn_s_waitcnt(0x0f76);
    else if (n == 5) __builtin_amdgcn_s_waitcnt(0x0f75);
    else if (n == 4) __builtin_amdgcn_s_waitcnt(0x0f74);
    else if (n == 3) __builtin_amdgcn_s_waitcnt(0x0f73);
    else if (n == 2) __builtin_amdgcn_s_waitcnt(0x0f72);
    else if (n == 1) __builtin_amdgcn_s_waitcnt(0x0f71);
    else __builtin_amdgcn_s_waitcnt(0x0f70);
}

// ---- register-direct epilogue (bf16 results) of the large-tile NT kernels: the wave owns the 128 x 64 patch at (mbase, nbase) ----
// ---- register-direct epilogue (bf16 results) ----
// acc[i][j][r] = C[m0 + wm*128 + j*16 + fr][n0 + wn*64 + i*16 + 4*fg + r]: a lane owns 4 consecutive columns of one row per
// (i, j) and the four lanes fr, fr+16, fr+32, fr+48 own 16 consecutive ones.  The 8-byte bf16 pieces are widened first: one
// v_permlane16_swap between the lanes 16 apart trades the second half of column group i against the first half of group
// i+1, after which a lane holds 8 consecutive columns = one 16-byte store, in 64-byte runs per row; the saved bf16
// activation of the dGELU / dReLU epilogues is read the same way.  No LDS, no barriers: the LDS-staged form below takes 6 us
// of a 27 us K = 768 tile (four passes, eight barriers, half the waves idle in each); measured on the workload's shapes the
// direct form is 5-13 % faster per launch for bf16 results.  fp32 results keep the staged form: its 1 KB runs per row beat
// 64-byte (direct) and 256-byte (per-wave patches) runs by 7-20 % on the HBM-bound shapes.  Bit 32768 of UENC_GEMM_VARIANT
// selects the staged form everywhere (A/B).
// NI = column groups of 16 the wave owns (4: the 64-wide patch; 3: the 48-wide patch of the 256 x 192 tile -- its third group has no
// partner to trade halves with and moves as 8-byte pieces: a third of the tile's bytes in 32-byte runs).
template <int EPI, int NI = 4>
__device__ __forceinline__ void nt_epilogue_direct(const GemmNT& p, f32x4 (&acc)[4][8], int mbase, int nbase, int fr, int fg, bool lead) {
    static_assert(NI == 4 || NI == 3, "column groups per wave");
    constexpr int NP = NI / 2;                          // pairs of column groups stored as 16-byte pieces
    constexpr bool ODD = (NI & 1) != 0;
    const int mrow = mbase + fr, ncol = nbase + 4 * fg;
    const int nodd = ncol + (NI - 1) * 16;              // this lane's 4 columns of the unpaired group
    float bv4[4][4];
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int n = ncol + i * 16;
        float4 b = make_float4(0.f, 0.f, 0.f, 0.f);
        if (p.bias != nullptr && n < p.N && lead) b = *(const float4*)(p.bias + n);
        bv4[i][0] = b.x; bv4[i][1] = b.y; bv4[i][2] = b.z; bv4[i][3] = b.w;
    }
    // column at which this lane's 16-byte bf16 store of pair pr (column groups 2 pr, 2 pr + 1) starts
    const int odd = fg & 1;
    const int n16[2] = {nbase + (0 + odd) * 16 + 4 * (fg - odd), nbase + (2 + odd) * 16 + 4 * (fg - odd)};
    // The saved activation / residual of ALL the wave's rows is requested before the first result store.  A wave's vector-memory
    // operations retire in order, and the compiler may not move these loads above the stores (C and aux may alias as far as it knows):
    // inside the row loop every row's load waited behind the previous row's stores -- eight load -> store round trips per tile,
    // 64 us of a 124 us N = 3072 dGELU launch against 32 us for the same bytes streamed.  Four rows per batch = two round trips;
    // all eight at once spilled (bit 1048576 of UENC_GEMM_VARIANT restores the per-row loads, A/B).
    constexpr bool AUX16 = (EPI == EPI_MUL_DGELU || EPI == EPI_MUL_DRELU);
    const bool hoist = !(p.variant & 1048576);
    u32x4 xa[AUX16 ? 4 : 1][2];                         // saved bf16 activation: four rows at a time (32 registers)
    u32x2 xo[AUX16 && ODD ? 4 : 1];                     // ... of the unpaired group
    float4 ra[EPI == EPI_RESIDUAL ? 2 : 1][4];          // fp32 residual: two rows at a time (32 registers)
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int m = mrow + j * 16;
        const bool row_ok = m < p.M;
        if (AUX16 && hoist && (j & 3) == 0) {
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                const int mm = mrow + (j + jj) * 16;
#pragma unroll
                for (int pr = 0; pr < NP; ++pr) {
                    xa[jj][pr] = (u32x4){0u, 0u, 0u, 0u};
                    if (mm < p.M && n16[pr] < p.N) xa[jj][pr] = *(const u32x4*)((const bf16*)p.aux + (long)mm * p.ldaux + n16[pr]);
                }
                if (ODD) {
                    xo[AUX16 && ODD ? jj : 0] = (u32x2){0u, 0u};
                    if (mm < p.M && nodd < p.N) xo[AUX16 && ODD ? jj : 0] = *(const u32x2*)((const bf16*)p.aux + (long)mm * p.ldaux + nodd);
                }
            }
        }
        if (EPI == EPI_RESIDUAL && hoist && (j & 1) == 0) {
#pragma unroll
            for (int jj = 0; jj < 2; ++jj)
#pragma unroll
                for (int i = 0; i < NI; ++i) {
                    const int mm = mrow + (j + jj) * 16, n = ncol + i * 16;
                    ra[jj][i] = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (mm < p.M && n < p.N) ra[jj][i] = *(const float4*)((const float*)p.aux + (long)mm * p.ldaux + n);
                }
        }
        float v[4][4];
        const float al = nt_alpha(p, min(m, p.M - 1));
#pragma unroll
        for (int i = 0; i < NI; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) v[i][r] = (acc[i][j][r] + bv4[i][r]) * al;
        u32x2 pre2[4];
        if (EPI == EPI_GELU) {
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                bf16x4 q;
#pragma unroll
                for (int r = 0; r < 4; ++r) { q[r] = (bf16)v[i][r]; v[i][r] = gelu_f(v[i][r]); }
                pre2[i] = *(const u32x2*)&q;
            }
        } else if (EPI == EPI_RELU) {
#pragma unroll
            for (int i = 0; i < NI; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) v[i][r] = fmaxf(v[i][r], 0.f);
        } else if (EPI == EPI_RESIDUAL) {
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                const int n = ncol + i * 16;
                if (hoist) {
                    const float4 r0 = ra[j & 1][i];
                    v[i][0] += r0.x; v[i][1] += r0.y; v[i][2] += r0.z; v[i][3] += r0.w;
                } else if (row_ok && n < p.N) {
                    const float4 r0 = *(const float4*)((const float*)p.aux + (long)m * p.ldaux + n);
                    v[i][0] += r0.x; v[i][1] += r0.y; v[i][2] += r0.z; v[i][3] += r0.w;
                }
            }
        } else if (EPI == EPI_MUL_DGELU || EPI == EPI_MUL_DRELU) {
            // the saved bf16 activation is read in the widened layout (16 bytes per lane) and brought back to the accumulator
            // layout by the same swap (an involution)
#pragma unroll
            for (int pr = 0; pr < NP; ++pr) {
                u32x4 x = {0u, 0u, 0u, 0u};
                if (hoist) x = xa[AUX16 ? (j & 3) : 0][pr];
                else if (row_ok && n16[pr] < p.N) x = *(const u32x4*)((const bf16*)p.aux + (long)m * p.ldaux + n16[pr]);
                const u32x2 lo = __builtin_amdgcn_permlane16_swap(x[0], x[2], false, false);
                const u32x2 hi = __builtin_amdgcn_permlane16_swap(x[1], x[3], false, false);
                const u32x2 sa = {lo[0], hi[0]}, sb = {lo[1], hi[1]};
                const bf16x4 s0 = *(const bf16x4*)&sa, s1 = *(const bf16x4*)&sb;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    v[2 * pr][r] = (EPI == EPI_MUL_DGELU) ? v[2 * pr][r] * dgelu_f((float)s0[r]) : (((float)s0[r] > 0.f) ? v[2 * pr][r] : 0.f);
                    v[2 * pr + 1][r] = (EPI == EPI_MUL_DGELU) ? v[2 * pr + 1][r] * dgelu_f((float)s1[r]) : (((float)s1[r] > 0.f) ? v[2 * pr + 1][r] : 0.f);
                }
            }
            if (ODD) {                                  // the unpaired group: already in the accumulator layout
                u32x2 x = {0u, 0u};
                if (hoist) x = xo[AUX16 && ODD ? (j & 3) : 0];
                else if (row_ok && nodd < p.N) x = *(const u32x2*)((const bf16*)p.aux + (long)m * p.ldaux + nodd);
                const bf16x4 s0 = *(const bf16x4*)&x;
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    v[NI - 1][r] = (EPI == EPI_MUL_DGELU) ? v[NI - 1][r] * dgelu_f((float)s0[r]) : (((float)s0[r] > 0.f) ? v[NI - 1][r] : 0.f);
            }
        }
        {
            u32x2 o2[4];
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                bf16x4 q;
#pragma unroll
                for (int r = 0; r < 4; ++r) q[r] = (bf16)v[i][r];
                o2[i] = *(const u32x2*)&q;
            }
            if (ODD && row_ok && nodd < p.N) *(u32x2*)((bf16*)p.C + (long)m * p.ldc + nodd) = o2[NI - 1];
#pragma unroll
            for (int pr = 0; pr < NP; ++pr) {
                const u32x2 lo = __builtin_amdgcn_permlane16_swap(o2[2 * pr][0], o2[2 * pr + 1][0], false, false);
                const u32x2 hi = __builtin_amdgcn_permlane16_swap(o2[2 * pr][1], o2[2 * pr + 1][1], false, false);
                if (row_ok && n16[pr] < p.N) {
                    const u32x4 o4 = {lo[0], hi[0], lo[1], hi[1]};
                    if (p.variant & 524288) __builtin_nontemporal_store(o4, (u32x4*)((bf16*)p.C + (long)m * p.ldc + n16[pr]));      // A/B: streaming stores
                    else *(u32x4*)((bf16*)p.C + (long)m * p.ldc + n16[pr]) = o4;
                }
            }
        }
        if (EPI == EPI_GELU && p.aux_out != nullptr) {
            if (ODD && row_ok && nodd < p.N) *(u32x2*)(p.aux_out + (long)m * p.ldaux_out + nodd) = pre2[NI - 1];
#pragma unroll
            for (int pr = 0; pr < NP; ++pr) {
                const u32x2 lo = __builtin_amdgcn_permlane16_swap(pre2[2 * pr][0], pre2[2 * pr + 1][0], false, false);
                const u32x2 hi = __builtin_amdgcn_permlane16_swap(pre2[2 * pr][1], pre2[2 * pr + 1][1], false, false);
                if (row_ok && n16[pr] < p.N) {
                    const u32x4 o4 = {lo[0], hi[0], lo[1], hi[1]};
                    if (p.variant & 524288) __builtin_nontemporal_store(o4, (u32x4*)(p.aux_out + (long)m * p.ldaux_out + n16[pr]));
                    else *(u32x4*)(p.aux_out + (long)m * p.ldaux_out + n16[pr]) = o4;
                }
            }
        }
    }
}

// ---- LDS-staged epilogue (fp32 results; bf16 under UENC_GEMM_VARIANT bit 32768): passes of 64 rows through a padded fp32 tile [64][260]
// in LDS, then 8 consecutive columns per thread: 1 KB runs per row.  NWM = row groups of 128 the workgroup owns (wave group wm each). ----
// NI = column groups of 16 per wave (4, or 3 for the 192-wide tile: the tile then spans 4 x 48 columns and the threads of the last 64 idle).
template <int EPI, int OUT_F32, int NTHREADS, int NWM, int NI = 4, bool LN = false>
__device__ __forceinline__ void nt_epilogue_staged(const GemmNT& p, f32x4 (&acc)[4][8], unsigned char* smem, int m0, int n0, int wm, int wn,
                                                   int t, int fr, int fg, bool lead, bool skip_stores) {
    float* T = (float*)smem;
    constexpr int LDT = 260;
    const int erow = t >> 5, ecol = (t & 31) * 8;
    constexpr int RPI = NTHREADS / 32;            // rows one iteration of the workgroup covers
    constexpr int NIT = 64 / RPI;
    const bool hoist = !(p.variant & 1048576) && !LN;       // (the fused-LayerNorm form loads its residual row by row)
    const int n = n0 + ecol;
    const bool ncol_ok = n < p.N && ecol < NI * 64;
    const bool full8 = (n + 8 <= p.N);
    constexpr bool ln = LN && EPI == EPI_RESIDUAL && OUT_F32;      // own instantiation (the launcher guarantees one column tile, N % 8 == 0)
    float lgam[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, lbet[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (ln && ncol_ok) {
        const float4 g0 = *(const float4*)(p.ln_gamma + n), g1 = *(const float4*)(p.ln_gamma + n + 4);
        const float4 b0 = *(const float4*)(p.ln_beta + n), b1 = *(const float4*)(p.ln_beta + n + 4);
        lgam[0] = g0.x; lgam[1] = g0.y; lgam[2] = g0.z; lgam[3] = g0.w; lgam[4] = g1.x; lgam[5] = g1.y; lgam[6] = g1.z; lgam[7] = g1.w;
        lbet[0] = b0.x; lbet[1] = b0.y; lbet[2] = b0.z; lbet[3] = b0.w; lbet[4] = b1.x; lbet[5] = b1.y; lbet[6] = b1.z; lbet[7] = b1.w;
    }
    float bv[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (p.bias != nullptr && ncol_ok && lead) {
        const float4 b0 = *(const float4*)(p.bias + n);
        bv[0] = b0.x; bv[1] = b0.y; bv[2] = b0.z; bv[3] = b0.w;
        if (full8) { const float4 b1 = *(const float4*)(p.bias + n + 4); bv[4] = b1.x; bv[5] = b1.y; bv[6] = b1.z; bv[7] = b1.w; }
    }
#pragma unroll
    for (int ps = 0; ps < 2 * NWM; ++ps) {      // unrolled: acc[][] must keep compile-time indices (else it lives in scratch)
        __syncthreads();
        // the pass's fp32 residual rows are requested here, ahead of the LDS transpose and of the previous pass's result stores in this
        // wave's in-order memory queue (each (row, 8 columns) piece belongs to exactly one thread and iteration, so the order of
        // these loads against other rows' stores does not matter even when aux aliases C); inside the row loop each load waited
        // for the stores before it
        float4 rres[EPI == EPI_RESIDUAL ? 4 : 1][2];              // four rows per batch (a 256-thread workgroup has eight per pass: two batches)
        auto fetch_res = [&](int it0) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int m = m0 + ps * 64 + (it0 + k) * RPI + erow;
                rres[EPI == EPI_RESIDUAL ? k : 0][0] = rres[EPI == EPI_RESIDUAL ? k : 0][1] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (m < p.M) {
                    const float* rp = (const float*)p.aux + (long)m * p.ldaux + n;
                    rres[EPI == EPI_RESIDUAL ? k : 0][0] = *(const float4*)rp;
                    if (full8) rres[EPI == EPI_RESIDUAL ? k : 0][1] = *(const float4*)(rp + 4);
                }
            }
        };
        if (EPI == EPI_RESIDUAL && hoist && ncol_ok) fetch_res(0);
        if (wm == (ps >> 1)) {
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                const int j = (ps & 1) * 4 + jj;
#pragma unroll
                for (int i = 0; i < NI; ++i) *(f32x4*)(T + (jj * 16 + fr) * LDT + wn * (NI * 16) + i * 16 + 4 * fg) = acc[i][j];
            }
        }
        __syncthreads();
        if (!ncol_ok && !ln) continue;
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int rl = it * RPI + erow;
            const int m = m0 + ps * 64 + rl;
            if (EPI == EPI_RESIDUAL && hoist && it == 4) fetch_res(4);
            if (m >= p.M) continue;
            if (ln) {
                // fused LayerNorm: the 32 lanes of this half-wave hold the row (lanes past N carry zeros and only take part in the sums)
                float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                if (ncol_ok) {
                    const f32x4 a0 = *(const f32x4*)(T + rl * LDT + ecol), a1 = *(const f32x4*)(T + rl * LDT + ecol + 4);
                    const float al = nt_alpha(p, m);
#pragma unroll
                    for (int r = 0; r < 4; ++r) { v[r] = (a0[r] + bv[r]) * al; v[4 + r] = (a1[r] + bv[4 + r]) * al; }
                    const float* rp = (const float*)p.aux + (long)m * p.ldaux + n;
                    const float4 r0 = *(const float4*)rp, r1 = *(const float4*)(rp + 4);
                    v[0] += r0.x; v[1] += r0.y; v[2] += r0.z; v[3] += r0.w; v[4] += r1.x; v[5] += r1.y; v[6] += r1.z; v[7] += r1.w;
                    float* c = (float*)p.C + (long)m * p.ldc + n;
                    *(float4*)c = make_float4(v[0], v[1], v[2], v[3]);
                    *(float4*)(c + 4) = make_float4(v[4], v[5], v[6], v[7]);
                }
                const float invn = 1.0f / (float)p.N;
                float s1 = ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
#pragma unroll
                for (int o = 1; o < 32; o <<= 1) s1 += __shfl_xor(s1, o);
                const float mean = s1 * invn;
                float d[8], s2 = 0.f;
#pragma unroll
                for (int r = 0; r < 8; ++r) { d[r] = ncol_ok ? v[r] - mean : 0.f; s2 += d[r] * d[r]; }
#pragma unroll
                for (int o = 1; o < 32; o <<= 1) s2 += __shfl_xor(s2, o);
                const float rstd = rsqrtf(s2 * invn + p.ln_eps);
                if (ncol_ok) {
                    float y[8];
#pragma unroll
                    for (int r = 0; r < 8; ++r) y[r] = d[r] * rstd * lgam[r] + lbet[r];
                    if (p.ln_y32 != nullptr) {
                        float* yp = p.ln_y32 + (long)m * p.N + n;
                        *(float4*)yp = make_float4(y[0], y[1], y[2], y[3]);
                        *(float4*)(yp + 4) = make_float4(y[4], y[5], y[6], y[7]);
                    }
                    if (p.ln_y16 != nullptr) {
                        bf16x8 o8;
#pragma unroll
                        for (int r = 0; r < 8; ++r) o8[r] = (bf16)y[r];
                        *(bf16x8*)(p.ln_y16 + (long)m * p.N + n) = o8;
                    }
                }
                if (ecol == 0 && p.ln_stats != nullptr) p.ln_stats[m] = make_float2(mean, rstd);
                continue;
            }
            const f32x4 a0 = *(const f32x4*)(T + rl * LDT + ecol), a1 = *(const f32x4*)(T + rl * LDT + ecol + 4);
            float v[8];
            const float al = nt_alpha(p, m);
#pragma unroll
            for (int r = 0; r < 4; ++r) { v[r] = (a0[r] + bv[r]) * al; v[4 + r] = (a1[r] + bv[4 + r]) * al; }
            if (skip_stores) {
                if (EPI == EPI_GELU) {
#pragma unroll
                    for (int r = 0; r < 8; ++r) v[r] = gelu_f(v[r]);
                }
#pragma unroll
                for (int r = 0; r < 8; ++r) asm volatile("" :: "v"(v[r]));
                continue;
            }
            if (EPI == EPI_GELU) {
                if (p.aux_out != nullptr) {
                    bf16x8 pre;
#pragma unroll
                    for (int r = 0; r < 8; ++r) pre[r] = (bf16)v[r];
                    bf16* dst = p.aux_out + (long)m * p.ldaux_out + n;
                    if (full8) *(bf16x8*)dst = pre;
                    else { bf16x4 q4; q4[0] = pre[0]; q4[1] = pre[1]; q4[2] = pre[2]; q4[3] = pre[3]; *(bf16x4*)dst = q4; }
                }
#pragma unroll
                for (int r = 0; r < 8; ++r) v[r] = gelu_f(v[r]);
            } else if (EPI == EPI_RELU) {
#pragma unroll
                for (int r = 0; r < 8; ++r) v[r] = fmaxf(v[r], 0.f);
            } else if (EPI == EPI_RESIDUAL) {
                if (hoist) {
                    const float4 r0 = rres[EPI == EPI_RESIDUAL ? (it & 3) : 0][0], r1 = rres[EPI == EPI_RESIDUAL ? (it & 3) : 0][1];
                    v[0] += r0.x; v[1] += r0.y; v[2] += r0.z; v[3] += r0.w;
                    v[4] += r1.x; v[5] += r1.y; v[6] += r1.z; v[7] += r1.w;
                } else {
                    const float* rp = (const float*)p.aux + (long)m * p.ldaux + n;
                    const float4 r0 = *(const float4*)rp;
                    v[0] += r0.x; v[1] += r0.y; v[2] += r0.z; v[3] += r0.w;
                    if (full8) { const float4 r1 = *(const float4*)(rp + 4); v[4] += r1.x; v[5] += r1.y; v[6] += r1.z; v[7] += r1.w; }
                }
            } else if (EPI == EPI_MUL_DGELU || EPI == EPI_MUL_DRELU) {
                const bf16* ap = (const bf16*)p.aux + (long)m * p.ldaux + n;
                bf16x8 sv;
                if (full8) sv = *(const bf16x8*)ap;
                else {
                    const bf16x4 q4 = *(const bf16x4*)ap;
#pragma unroll
                    for (int r = 0; r < 4; ++r) { sv[r] = q4[r]; sv[4 + r] = (bf16)0.f; }
                }
#pragma unroll
                for (int r = 0; r < 8; ++r)
                    v[r] = (EPI == EPI_MUL_DGELU) ? v[r] * dgelu_f((float)sv[r]) : (((float)sv[r] > 0.f) ? v[r] : 0.f);
            }
            if (OUT_F32) {
                float* c = (float*)p.C + (long)m * p.ldc + n;
                if (EPI == EPI_NONE && p.atomic) {
#pragma unroll
                    for (int r = 0; r < 8; ++r)
                        if (r < 4 || full8) atomicAdd(c + r, v[r]);
                } else {
                    *(float4*)c = make_float4(v[0], v[1], v[2], v[3]);
                    if (full8) *(float4*)(c + 4) = make_float4(v[4], v[5], v[6], v[7]);
                }
            } else {
                bf16x8 o;
#pragma unroll
                for (int r = 0; r < 8; ++r) o[r] = (bf16)v[r];
                bf16* c = (bf16*)p.C + (long)m * p.ldc + n;
                if (full8) *(bf16x8*)c = o;
                else { bf16x4 q4; q4[0] = o[0]; q4[1] = o[1]; q4[2] = o[2]; q4[3] = o[3]; *(bf16x4*)c = q4; }
            }
        }
    }
}

// BNT = 192: the 256 x 192 tile (PIPE = 1 only).  A wave then owns 128 x 48 = three column groups: B0 keeps the first two, B1 is the third
// alone (a 64-row, 8 KB half-tile: ONE DMA instruction per thread, and quadrants of 16 / 8 / 8 / 16 MFMAs).  For the shapes whose
// 256-wide tiling leaves CUs idle or columns empty: N = 768 at M = 16384 is 192 tiles of 256 x 256 on 256 CUs but 256 tiles of
// 256 x 192; N = 2304 is 2.25 rounds of tiles against 3 rounds of 3/4 the work; N = 384 / 576 / 192 waste a quarter of their last column.
template <int EPI, int OUT_F32, int PIPE, int BNT = BN2, bool LN = false>
__global__ __launch_bounds__(T2) void gemm_nt256_kernel(GemmNT p) {
    static_assert(BNT == 256 || (BNT == 192 && PIPE == 1), "tile width");
    constexpr int NI = BNT / 64;          // column groups of 16 per wave
    constexpr int NB1 = NI - 2;           // ... of them in the B1 half-tile
    constexpr int BNW = BNT / 4;          // columns per wave
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];     // 2 stages x (A 32 KB + W 32 KB)
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int wm = wave >> 2, wn = wave & 3;
    // Persistent form (p.persist, launches with more tiles than CUs): a workgroup walks the tile positions blockIdx.x,
    // blockIdx.x + gridDim.x, ... (the positions in-order dispatch of one workgroup per tile would have given this CU; gridDim.x is a
    // multiple of 8, so a workgroup stays on the XCD chunk xcd_remap gives its slot).  The next tile's first six half-tile DMAs are
    // issued BEFORE the epilogue of the current tile when that epilogue leaves the ring alone (bf16 results), else right after it:
    // their latency, the workgroup turn-around and the drain of the result stores then overlap work instead of following it.
    const int ntiles_all = p.tiles_m * p.tiles_n;
    int pos = blockIdx.x;
    bool prefetched = false;
    const bf16* src[4][2];              // [A0, B0, B1, A1][q]: DMA instruction q of this thread (row L >> 3, slot L & 7, L = q * 512 + t)
    for (;;) {
    const int tile = xcd_remap(pos, ntiles_all);
    const int mt = tile / p.tiles_n, nt = tile - mt * p.tiles_n;
    const int m0 = mt * BM2, n0 = nt * BNT;
    const int pos_next = pos + (int)gridDim.x;
    const bool has_next = p.persist && pos_next < ntiles_all;
    // split-K (EPI_NONE, fp32 C, pre-zeroed or accumulated into): blockIdx.y owns k-tiles [kt0, kt0 + nkt), adds its tile atomically
    // (or, with part_stride > 0, stores it as partial sum number blockIdx.y: summed by the caller)
    const int kt0 = blockIdx.y * (p.klen / BK);
    const int nkt = min(p.K / BK - kt0, p.klen / BK);
    if (OUT_F32 && p.part_stride > 0) p.C = (float*)p.C + (long)blockIdx.y * p.part_stride;

    f32x4 acc[4][8];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    typedef __attribute__((address_space(3))) void lds_void;
    const int fr = lane & 15, fg = lane >> 4;
    if (PIPE) {
        // ---- staggered quadrant pipeline ----
        // A k-tile buffer (64 KB, two of them) holds four 16 KB HALF-tiles, 128 rows x 128 B each: A0 / A1 = the rows of every wave's
        // first / second 64 output rows, B0 / B1 = the W rows of every wave's first / second 32 output columns.  A k-tile is four
        // phases, one output quadrant (64 x 32 per wave, 16 MFMAs) each, in the order (A0,B0) (A0,B1) (A1,B1) (A1,B0); a phase is
        //     ds_read the operands the quadrant still lacks | issue ONE half-tile DMA | counted vmcnt | s_barrier | 16 MFMAs | s_barrier
        // and the two wave groups (wm = 0 / 1: the two waves of every SIMD) run one barrier apart, so that one group's MFMAs cover the
        // other's LDS reads and DMA issue.  The DMA stream runs 6 half-tiles ahead of the reads (half-tile u = 4 T + {A0,B0,B1,A1}; phase
        // s issues u = s + 6 into the slot read last two or three phases ago) and `vmcnt(8)` after the issue leaves four of them in
        // flight: u <= s + 2 has landed in this wave, and in all waves after the next barrier pair, i.e. for the reads of phase s + 1.
        auto set_src = [&](int mm0, int nn0) {
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int lr = q * 64 + (t >> 3), slot = t & 7;
                const int chunk = slot ^ ((lr >> 1) & 7);
                const int ra = (lr >> 6) * 128 + (lr & 63), rb = (lr >> 5) * BNW + (lr & 31);
                const int rb1 = NB1 == 2 ? rb + 32 : (lr >> 4) * BNW + 32 + (lr & 15);      // (192: B1 has 64 rows, 16 per wave; only q = 0 is issued)
                src[0][q] = (const bf16*)p.A + (long)min(mm0 + ra, p.M - 1) * p.lda + chunk * 8 + (long)kt0 * BK;
                src[3][q] = (const bf16*)p.A + (long)min(mm0 + ra + 64, p.M - 1) * p.lda + chunk * 8 + (long)kt0 * BK;
                src[1][q] = p.W + (long)min(nn0 + rb, p.N - 1) * p.ldw + chunk * 8 + (long)kt0 * BK;
                src[2][q] = p.W + (long)min(nn0 + rb1, p.N - 1) * p.ldw + chunk * 8 + (long)kt0 * BK;
            }
        };
        if (!prefetched) set_src(m0, n0);
        // LDS offsets of the half-tiles inside a k-tile buffer, by sequence position o = 0..3 (A0, B0, B1, A1)
        const int U = 4 * nkt;
        auto issue = [&](int o, int tile) {          // o compile-time after inlining
            unsigned char* dst = smem + (tile & 1) * 65536 + (o == 0 ? 0 : o == 1 ? 32768 : o == 2 ? 49152 : 16384) + wave * 1024;
#pragma unroll
            for (int q = 0; q < 2; ++q)
                if (q == 0 || o != 2 || NB1 == 2)
                    __builtin_amdgcn_global_load_lds(src[o][q] + (long)tile * BK, (lds_void*)(dst + q * 8192), 16, 0, 0);
        };
        // DMA instructions of this thread still in flight when half-tiles first .. min(first + 3, U - 1) are (2 each; 1 for the 64-row B1)
        auto inflight = [&](int first) {
            if (NB1 == 2) return 2 * min(max(U - first, 0), 4);
            int n = 0;
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (first + k < U) n += (((first + k) & 3) == 2) ? 1 : 2;
            return n;
        };
        // prologue: half-tiles 0..5 (k-tile 0 and A0, B0 of k-tile 1); A0, B0 of k-tile 0 must have landed everywhere before phase 0
        // (the 192-wide tile consumes a k-tile as A0 B0 | A1 | B1 and issues in that order: see its loop below)
        auto prologue = [&]() {
            if (BNT == 192) { issue(0, 0); issue(1, 0); issue(3, 0); issue(2, 0); }
            else { issue(0, 0); issue(1, 0); issue(2, 0); issue(3, 0); }
            if (U > 4) { issue(0, 1); issue(1, 1); }
        };
        if (!prefetched) prologue();
        // (a prefetched prologue is followed by the previous tile's result stores: they are younger, so this wait also covers them
        // up to the last 8 -- conservative, never early)
        if (BNT == 192) wait_vmcnt_upto8(3 + (U > 4 ? 4 : 0));  // (A1, B1 of k-tile 0 and A0, B0 of k-tile 1 may still be in flight)
        else wait_vmcnt_upto8(inflight(2));                    // (half-tiles 2 .. 5 may still be in flight)
        __builtin_amdgcn_s_barrier();
        if (wm == 1) __builtin_amdgcn_s_barrier();             // the stagger: group 1 runs one barrier behind group 0

        const int aoff = lds_off(wm * 64 + fr, fg), boff = lds_off(wn * 32 + fr, fg);      // row + 16 j keeps (row >> 1) & 7 -> + 2048 j
        const int boff1 = NB1 == 2 ? boff : lds_off(wn * 16 + fr, fg);
        if constexpr (BNT == 192) {
        // ---- 256 x 192: THREE phases of 16 MFMAs per k-tile (the four-quadrant order would give 16 / 8 / 8 / 16, and an 8-MFMA phase
        // is shorter than the other wave group's LDS reads it has to cover: measured 35-43 % slower per flop than the 256-wide tile) ----
        //     a: read B0 (2 column groups), A0 | issue A1 of k-tile T+1       | (A0, B0): 16 MFMAs
        //     b: read A1                        | issue B1 of k-tile T+1       | (A1, B0): 16 MFMAs
        //     c: read B1 (1 column group)       | issue A0, B0 of k-tile T+2   | (A0 and A1, B1): 16 MFMAs   (A0 stays in registers)
        // Every DMA goes into a slot last read two phases before (the other wave group runs one barrier behind and may still have that
        // slot's ds_reads in flight one phase after); the stream order per k-tile is A0 B0 A1 B1 = the order of use, each piece one
        // k-tile (three phases) ahead of its reads.  The wait of a phase covers the NEXT phase's operands in this wave; the barrier
        // pair that follows covers the other waves'.  After the required piece at most 7 instructions are younger (2 + 2 + 2 + 1).
        bf16x8 af0[4][2], af1[4][2], b0[2][2], b1[1][2];
        for (int T = 0; T < nkt; ++T) {
            const unsigned char* buf = smem + (T & 1) * 65536;
            const bool n1 = T + 1 < nkt, n2 = T + 2 < nkt;
            // phase a
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) b0[i][ks] = *(const bf16x8*)(buf + 32768 + i * 2048 + (boff ^ (ks << 6)));
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) af0[j][ks] = *(const bf16x8*)(buf + j * 2048 + (aoff ^ (ks << 6)));
            if (n1) issue(3, T + 1);
            wait_vmcnt_upto8(n1 ? 7 : 1);                      // A1 of this k-tile has landed (B1 and the next k-tile may be in flight)
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[i][j] = mfma16(b0[i][ks], af0[j][ks], acc[i][j]);
            __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            // phase b
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) af1[j][ks] = *(const bf16x8*)(buf + 16384 + j * 2048 + (aoff ^ (ks << 6)));
            if (n1) issue(2, T + 1);
            wait_vmcnt_upto8(n1 ? 7 : 0);                      // B1 of this k-tile has landed
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[i][4 + j] = mfma16(b0[i][ks], af1[j][ks], acc[i][4 + j]);
            __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            // phase c
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) b1[0][ks] = *(const bf16x8*)(buf + 49152 + (boff1 ^ (ks << 6)));
            if (n2) { issue(0, T + 2); issue(1, T + 2); }
            wait_vmcnt_upto8((n1 ? 3 : 0) + (n2 ? 4 : 0));     // A0, B0 of the next k-tile have landed
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[2][j] = mfma16(b1[0][ks], af0[j][ks], acc[2][j]);
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[2][4 + j] = mfma16(b1[0][ks], af1[j][ks], acc[2][4 + j]);
            }
            __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
        }
        } else {
        bf16x8 af[4][2], b0[2][2], b1[NB1][2];
        for (int T = 0; T < nkt; ++T) {
            const unsigned char* buf = smem + (T & 1) * 65536;
            const int s = 4 * T;
            // phase 1: A0, B0 -> quadrant (0, 0)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) b0[i][ks] = *(const bf16x8*)(buf + 32768 + i * 2048 + (boff ^ (ks << 6)));
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) af[j][ks] = *(const bf16x8*)(buf + j * 2048 + (aoff ^ (ks << 6)));
            if (s + 6 < U) issue(2, T + 1);
            wait_vmcnt_upto8(inflight(s + 3));
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[i][j] = mfma16(b0[i][ks], af[j][ks], acc[i][j]);
            __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            // phase 2: B1 -> quadrant (0, 1)
#pragma unroll
            for (int i = 0; i < NB1; ++i)
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) b1[i][ks] = *(const bf16x8*)(buf + 49152 + i * 2048 + (boff1 ^ (ks << 6)));
            if (s + 7 < U) issue(3, T + 1);
            wait_vmcnt_upto8(inflight(s + 4));
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int i = 0; i < NB1; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[2 + i][j] = mfma16(b1[i][ks], af[j][ks], acc[2 + i][j]);
            __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            // phase 3: A1 -> quadrant (1, 1)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) af[j][ks] = *(const bf16x8*)(buf + 16384 + j * 2048 + (aoff ^ (ks << 6)));
            if (s + 8 < U) issue(0, T + 2);
            wait_vmcnt_upto8(inflight(s + 5));
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int i = 0; i < NB1; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[2 + i][4 + j] = mfma16(b1[i][ks], af[j][ks], acc[2 + i][4 + j]);
            __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            // phase 4: (B0 still in registers) -> quadrant (1, 0)
            if (s + 9 < U) issue(1, T + 2);
            wait_vmcnt_upto8(inflight(s + 6));
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[i][4 + j] = mfma16(b0[i][ks], af[j][ks], acc[i][4 + j]);
            __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
        }
        }
        if (wm == 0) __builtin_amdgcn_s_barrier();             // group 0 waits for group 1's last phase
        prefetched = false;
        if (has_next && !OUT_F32 && !(p.variant & 32768)) {     // the register-direct epilogue does not touch LDS: start the next tile's loads now
            const int tile2 = xcd_remap(pos_next, ntiles_all);
            const int mt2 = tile2 / p.tiles_n;
            set_src(mt2 * BM2, (tile2 - mt2 * p.tiles_n) * BNT);
            __builtin_amdgcn_sched_barrier(0);
            prologue();
            prefetched = true;
        }
    } else {
        // DMA geometry: instruction q of a thread fills LDS chunk index L = q * 512 + t  (row L >> 3, slot L & 7)
        const bf16* asrc[4];
        const bf16* wsrc[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int row = q * 64 + (t >> 3), slot = t & 7;
            const int chunk = slot ^ ((row >> 1) & 7);
            asrc[q] = (const bf16*)p.A + (long)min(m0 + row, p.M - 1) * p.lda + chunk * 8 + (long)kt0 * BK;
            wsrc[q] = p.W + (long)min(n0 + row, p.N - 1) * p.ldw + chunk * 8 + (long)kt0 * BK;
        }
        auto issue = [&](int kt, int stage) {
            unsigned char* As = smem + stage * 65536;
            unsigned char* Ws = As + 32768;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                __builtin_amdgcn_global_load_lds(asrc[q] + kt * BK, (lds_void*)(As + (q * 512 + wave * 64) * 16), 16, 0, 0);
                __builtin_amdgcn_global_load_lds(wsrc[q] + kt * BK, (lds_void*)(Ws + (q * 512 + wave * 64) * 16), 16, 0, 0);
            }
        };

        issue(0, 0);
        for (int kt = 0; kt < nkt; ++kt) {
            __syncthreads();     // waits for this wave's DMA (vmcnt 0), then everyone: tile kt landed, stage (kt+1)&1 is free
            if (kt + 1 < nkt) issue(kt + 1, (kt + 1) & 1);
            const unsigned char* As = smem + (kt & 1) * 65536;
            const unsigned char* Ws = As + 32768;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                bf16x8 wf[4], xf[8];
#pragma unroll
                for (int i = 0; i < 4; ++i) wf[i] = *(const bf16x8*)(Ws + lds_off(wn * 64 + i * 16 + fr, ks * 4 + fg));
#pragma unroll
                for (int j = 0; j < 8; ++j) xf[j] = *(const bf16x8*)(As + lds_off(wm * 128 + j * 16 + fr, ks * 4 + fg));
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 8; ++j) acc[i][j] = mfma16(wf[i], xf[j], acc[i][j]);
            }
        }

    }

    // epilogue: four passes of 64 rows through a padded fp32 LDS tile [64][260]; then 8 columns per thread
    if (p.variant & 16384) {                 // timing experiment: no epilogue at all (keeps the accumulators alive)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j) asm volatile("" :: "v"(acc[i][j]));
        if (!has_next) return;
        pos = pos_next;
        continue;
    }
    if (!OUT_F32 && !(p.variant & 32768)) {
        nt_epilogue_direct<EPI, NI>(p, acc, m0 + wm * 128, n0 + wn * BNW, fr, fg, blockIdx.y == 0);
        if (!has_next) return;
        pos = pos_next;
        continue;
    }
    nt_epilogue_staged<EPI, OUT_F32, T2, 2, NI, LN>(p, acc, smem, m0, n0, wm, wn, t, fr, fg, blockIdx.y == 0, (p.variant & 8192) != 0);
    if (!has_next) return;
    __syncthreads();          // the staged epilogue's last reads of the LDS tile precede the next tile's DMA writes
    pos = pos_next;
    }
}

// ---------------------------------------------------------------------------------------------
// Half-height variant with TWO workgroups per CU: 128 x 256 x 32 tile, 4 waves (1 x 4), each wave the same 128(m) x 64(n) patch of
// 8 x 4 MFMA tiles as in the 256-tile kernel, 72 KB of LDS (ring of three 24 KB k-steps: A 128 rows + W 256 rows of 64 bytes).
// Why: the 256-tile kernel owns its CU alone, and vector-memory operations of a wave retire in order -- a tile's result stores sit
// in front of the next tile's operand DMA in the same queue, so a CU alternates between "MFMA" and "drain stores / wait for loads".
// On the shapes whose epilogue moves as many bytes as the main loop computes (K <= 384, fp32 residual / saved-activation epilogues,
// N >= 2304 with GELU) that serialisation is 30-50 % of the launch (profiles/r03_gemm_decompose.txt).  Two independent workgroups per
// CU have independent queues: one's epilogue stores and prologue loads run under the other's MFMAs, and the workgroup scheduler,
// not a persistent loop, refills a slot as soon as a tile retires.  Price: A and W panels travel L2 -> LDS 1.5 x as often per flop.
//   main loop, per 32-wide k-step s:   vmcnt(6): step s has landed (step s + 1 stays in flight) | s_barrier (everyone's step s is in LDS,
//   everyone has finished reading step s - 1) | DMA of step s + 2 into the buffer step s - 1 used | 12 ds_read_b128 | 32 MFMAs
// LDS rows are 64 bytes (four 16-byte chunks); chunk c of row r lives at slot c ^ perm[(r >> 2) & 3], perm = {0, 3, 2, 1}: the
// 16-lane groups a ds_read_b128 is serviced in ({0-3, 12-15, 20-27}, ...: MI355X_MICROARCH.md §LDS) then cover all 64 banks once.
// As in the 256-tile kernel the swizzle is applied to the per-lane SOURCE address of the lane-linear LDS-DMA.
// ---------------------------------------------------------------------------------------------
#define BM3 128
#define BN3 256
#define BK3 32
#define T3 256
#define STAGE3 ((BM3 + BN3) * BK3 * 2)      // 24576 bytes

__device__ __forceinline__ int lds_off32(int row, int chunk) {
    const int f = (4 - ((row >> 2) & 3)) & 3;                  // perm {0, 3, 2, 1}
    return row * 64 + ((chunk ^ f) << 4);
}

template <int EPI, int OUT_F32, bool LN = false>
__global__ __launch_bounds__(T3, 2) void gemm_nt128_kernel(GemmNT p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];     // 3 x (A 8 KB + W 16 KB)
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int wn = wave;
    const int ntiles_all = p.tiles_m * p.tiles_n;
    const int tile = xcd_remap(blockIdx.x, ntiles_all);
    const int mt = tile / p.tiles_n, nt = tile - mt * p.tiles_n;
    const int m0 = mt * BM3, n0 = nt * BN3;
    const int nks = p.K / BK3;

    f32x4 acc[4][8];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    typedef __attribute__((address_space(3))) void lds_void;
    const int fr = lane & 15, fg = lane >> 4;
    // DMA geometry: instruction q of a thread fills LDS chunk L = q * 256 + t of an image (row L >> 2, slot L & 3)
    const bf16* asrc[2];
    const bf16* wsrc[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int L = q * T3 + t, row = L >> 2, slot = L & 3;
        const int chunk = slot ^ ((4 - ((row >> 2) & 3)) & 3);
        if (q < 2) asrc[q] = (const bf16*)p.A + (long)min(m0 + row, p.M - 1) * p.lda + chunk * 8;
        wsrc[q] = p.W + (long)min(n0 + row, p.N - 1) * p.ldw + chunk * 8;
    }
    auto issue = [&](int ks, int buf) {
        unsigned char* As = smem + buf * STAGE3 + wave * 1024;
        unsigned char* Ws = As + BM3 * BK3 * 2;
#pragma unroll
        for (int q = 0; q < 2; ++q) __builtin_amdgcn_global_load_lds(asrc[q] + (long)ks * BK3, (lds_void*)(As + q * 4096), 16, 0, 0);
#pragma unroll
        for (int q = 0; q < 4; ++q) __builtin_amdgcn_global_load_lds(wsrc[q] + (long)ks * BK3, (lds_void*)(Ws + q * 4096), 16, 0, 0);
    };
    issue(0, 0);
    if (nks > 1) issue(1, 1);
    const int aoff = lds_off32(fr, fg), boff = lds_off32(wn * 64 + fr, fg);       // row + 16 j keeps (row >> 2) & 3 -> + 1024 j
    int buf = 0;
    for (int s = 0; s < nks; ++s) {
        if (s + 1 < nks) __builtin_amdgcn_s_waitcnt(0x0f76);      // vmcnt(6): step s landed, step s + 1 in flight
        else __builtin_amdgcn_s_waitcnt(0x0f70);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        if (s + 2 < nks) issue(s + 2, buf == 0 ? 2 : buf - 1);
        const unsigned char* As = smem + buf * STAGE3;
        const unsigned char* Ws = As + BM3 * BK3 * 2;
        bf16x8 wf[4], xf[8];
#pragma unroll
        for (int i = 0; i < 4; ++i) wf[i] = *(const bf16x8*)(Ws + i * 1024 + boff);
#pragma unroll
        for (int j = 0; j < 8; ++j) xf[j] = *(const bf16x8*)(As + j * 1024 + aoff);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[i][j] = mfma16(wf[i], xf[j], acc[i][j]);
        buf = buf == 2 ? 0 : buf + 1;
    }
    if (!OUT_F32 && !(p.variant & 32768)) {
        nt_epilogue_direct<EPI>(p, acc, m0, n0 + wn * 64, fr, fg, true);
        return;
    }
    __syncthreads();          // the staged epilogue reuses the ring
    nt_epilogue_staged<EPI, OUT_F32, T3, 1, 4, LN>(p, acc, smem, m0, n0, 0, wn, t, fr, fg, true, false);
}

template <int EPI, int OUT_F32, bool LN = false>
static int launch_nt128(GemmNT& p, hipStream_t stream) {
    p.tiles_m = (p.M + BM3 - 1) / BM3; p.tiles_n = (p.N + BN3 - 1) / BN3;
    static bool attr_set = false;      // per instantiation
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)gemm_nt128_kernel<EPI, OUT_F32, LN>, hipFuncAttributeMaxDynamicSharedMemorySize, 3 * STAGE3);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    p.persist = 0;
    hipLaunchKernelGGL((gemm_nt128_kernel<EPI, OUT_F32, LN>), dim3(p.tiles_m * p.tiles_n), dim3(T3), 3 * STAGE3, stream, p);
    return UENC_OK;
}

static int nt_cu_count() {
    static int ncu = 0;
    if (ncu == 0) {
        int dev = 0, v = 0;
        if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v >= 8) ncu = v / 8 * 8;
        else ncu = 256;
    }
    return ncu;
}

template <int EPI, int OUT_F32, int PIPE, int BNT = BN2, bool LN = false>
static int launch_nt256(GemmNT& p, hipStream_t stream) {
    p.tiles_m = (p.M + BM2 - 1) / BM2; p.tiles_n = (p.N + BNT - 1) / BNT;
    static bool attr_set = false;      // per instantiation
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)gemm_nt256_kernel<EPI, OUT_F32, PIPE, BNT, LN>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    // more tiles than CUs: one persistent workgroup per CU walks its tile positions (bit 65536 of UENC_GEMM_VARIANT: one workgroup per tile)
    const int ncu = nt_cu_count();
    const int ntiles = p.tiles_m * p.tiles_n;
    p.persist = (PIPE == 1 && p.splits == 1 && ntiles > ncu && !(p.variant & 65536)) ? 1 : 0;
    hipLaunchKernelGGL((gemm_nt256_kernel<EPI, OUT_F32, PIPE, BNT, LN>), dim3(p.persist ? ncu : ntiles, p.splits), dim3(T2), 131072, stream, p);
    return UENC_OK;
}

// 256 x 192 or 256 x 256 tiles?  Rounds of tiles over the CUs x cost per tile.  Measured (tools/gemm_nt192_ab.py ->
// profiles/r03_gemm_nt192_ab.txt): a 192-wide tile costs 0.85-0.92 of a 256-wide one, not 0.75 (its DMA stream runs one k-tile ahead
// instead of 1.5, and 22 instead of 24 ds_reads feed 48 instead of 64 MFMAs), so it is taken where it saves a round or a quarter-empty
// column: N = 768 / 2304 at M = 16384 (-8...-10 %), N = 384 (-8...-12 %), N = 4608 / 6144 at M = 4096 (-5...-15 %); neutral on the HBM-bound
// N = 192 / 576 launches of stage 1.  UENC_GEMM_VARIANT bit 4194304 forces the 192-wide tile wherever it is legal, bit 8388608 forbids it (A/B).
static bool nt192_wins(int M, int N, int variant) {
    if (variant & 8388608) return false;
    if (variant & 4194304) return true;
    const long ncu = nt_cu_count(), tm = (M + BM2 - 1) / BM2;
    const long r256 = (tm * ((N + 255) / 256) + ncu - 1) / ncu, r192 = (tm * ((N + 191) / 192) + ncu - 1) / ncu;
    return r192 * 90 < r256 * 100;
}

// Which of the two large-tile kernels takes a shape (both compute the same sums in the same k order: identical results).
// UENC_GEMM_VARIANT bit 131072 forces the two-workgroups-per-CU kernel, bit 262144 the one-workgroup kernel (A/B).
static bool nt128_wins(int M, int N, int K, int epilogue, int c_dtype, int variant) {
    if (variant & 262144) return false;
    if (variant & 131072) return true;
    // Measured per workload shape (tools/gemm_nt128_ab.py -> profiles/r03_gemm_nt128_ab.txt): the half-height kernel wins 3-15 % where a
    // single 256-wide column of tiles covers N and the contraction is short (the deformable encoder's K = 256 projections and its
    // 1024 -> 256 FFN GEMM with the fp32 residual epilogue, the decoder's key / value projections over the 1/4-resolution map): every
    // A row is read once and the launch is store-bound.  Everywhere else its 1.5 x L2 -> LDS traffic and 1.5 x DMA instructions per
    // MFMA cost 4-30 %.
    (void)epilogue;
    if (c_dtype == UENC_BF16 && M > 200000) return false;          // (262144, 256, 256) -> bf16: 6 % slower
    return N > 192 && N <= 288 && K <= 1024;
}

struct NtLn { const float* gamma; const float* beta; float* y32; void* y16; float* stats; float eps; };

static int gemm_nt_impl(const void* A, int a_dtype, long lda, const void* W, long ldw, void* C, int c_dtype, long ldc,
                        int M, int N, int K, const float* bias, int epilogue, const void* aux, long ldaux,
                        void* aux_out, long ldaux_out, float alpha, int splitk, int accumulate, int batch, long bsA, long bsW,
                        long bsC, hipStream_t stream, long part_stride = 0, const float* sample_scale = nullptr, int rows_per_sample = 0,
                        const NtLn* ln = nullptr) {
    UENC_CHECK_ARG(A && W && C && M > 0 && N > 0 && K > 0);
    UENC_CHECK_ARG(a_dtype == UENC_F32 || a_dtype == UENC_BF16);
    UENC_CHECK_ARG(c_dtype == UENC_F32 || c_dtype == UENC_BF16);
    UENC_CHECK_ARG(K % 8 == 0 && N % 4 == 0 && ldw % 8 == 0 && ldc % 4 == 0);
    UENC_CHECK_ARG(a_dtype == UENC_F32 ? (lda % 4 == 0) : (lda % 8 == 0));
    UENC_CHECK_ARG(((uintptr_t)A & 15) == 0 && ((uintptr_t)W & 15) == 0 && ((uintptr_t)C & 15) == 0);
    UENC_CHECK_ARG(epilogue >= EPI_NONE && epilogue <= EPI_MUL_DRELU);
    if (epilogue >= EPI_RESIDUAL) UENC_CHECK_ARG(aux != nullptr && ldaux % 4 == 0);
    if (splitk < 1) splitk = 1;
    if (splitk > 1 || accumulate) UENC_CHECK_ARG(c_dtype == UENC_F32 && epilogue == EPI_NONE);
    GemmNT p;
    { const char* e = getenv("UENC_GEMM_VARIANT"); p.variant = e ? atoi(e) : 0; }
    p.A = A; p.a_f32 = (a_dtype == UENC_F32); p.lda = lda;
    p.W = (const bf16*)W; p.ldw = ldw;
    p.C = C; p.ldc = ldc; p.bias = bias; p.aux = aux; p.ldaux = ldaux;
    p.aux_out = (bf16*)aux_out; p.ldaux_out = ldaux_out;
    p.M = M; p.N = N; p.K = K;
    p.tiles_m = (M + BM - 1) / BM; p.tiles_n = (N + BN - 1) / BN;
    const int kt = (K + BK - 1) / BK;
    if (splitk > kt) splitk = kt;
    p.klen = ((kt + splitk - 1) / splitk) * BK;
    splitk = (K + p.klen - 1) / p.klen;
    p.atomic = (splitk > 1 || accumulate) ? 1 : 0;
    p.part_stride = 0; p.splits = splitk;
    const bool partials = part_stride > 0;
    if (partials) {       // stored partial sums: plain epilogue per split, no atomics
        UENC_CHECK_ARG(c_dtype == UENC_F32 && epilogue == EPI_NONE && !accumulate && bias == nullptr && batch == 1 && part_stride % 4 == 0);
        p.atomic = 0; p.part_stride = part_stride;
    }
    p.alpha = alpha;
    p.sample_scale = nullptr; p.inv_rows = 0.f;
    if (sample_scale != nullptr) {        // per-sample multiplier of alpha (stored results only)
        UENC_CHECK_ARG(rows_per_sample > 0 && !p.atomic && !partials && batch == 1 && M < (1 << 22));
        p.sample_scale = sample_scale; p.inv_rows = 1.0f / (float)rows_per_sample;
    }
    p.bsA = bsA; p.bsW = bsW; p.bsC = bsC;
    p.ln_gamma = nullptr; p.ln_beta = nullptr; p.ln_y32 = nullptr; p.ln_y16 = nullptr; p.ln_stats = nullptr; p.ln_eps = 0.f;
    dim3 grid(p.tiles_m * p.tiles_n, splitk, batch), block(GEMM_THREADS);
    const bool prof = uenc_prof_on();
    // large-tile path: bf16 A, K a multiple of 64, no split-K, enough 256x256 tiles to fill most CUs
    const long tiles256 = (long)((M + 255) / 256) * ((N + 255) / 256);
    const int nmin = (p.variant & 64) ? 256 : 192;      // a 192-wide output still wins on the 256 tile: A is streamed once, not twice
    // (a contraction that is a multiple of 32 but not of 64 -- the 288 columns of the deformable encoder's offset / weight projection -- fits the
    // half-height kernel's 32-wide k-steps: it goes there when that kernel is the choice anyway, instead of to the register-staged one)
    const bool k32 = (K % BK != 0) && (K % BK3 == 0) && !p.atomic && !partials && !(p.variant & 33554432) &&
                     nt128_wins(M, N, K, epilogue, c_dtype, p.variant);
    const bool big = batch == 1 && a_dtype == UENC_BF16 && ((K % BK == 0) || k32) && K >= 128 && lda % 8 == 0 && !(p.variant & 2) && N >= nmin && N % 8 == 0 &&
                     ((p.atomic || partials) ? (epilogue == EPI_NONE && c_dtype == UENC_F32 && p.klen % BK == 0 && tiles256 >= 4 && tiles256 * splitk >= 64 &&
                                  !(p.variant & 16))     // (a single skinny tile measured faster on the 128x128 kernel)
                               : tiles256 >= 160);
    // too few 256 x 256 tiles for the one-workgroup-per-CU kernel, but >= 96 tiles of 128 x 256 (stage 4 of Swin-L: M = 4096, N = 1536: 192; the
    // key / value projections of the decoder's smaller pyramid levels): the half-height kernel instead of the register-staged 128 x 128 one
    // (bit 2097152 of UENC_GEMM_VARIANT: the old choice, A/B)
    const long htiles = (long)((M + 127) / 128) * ((N + 255) / 256);
    const bool mid = !big && batch == 1 && a_dtype == UENC_BF16 && (K % BK == 0) && K >= 128 && lda % 8 == 0 && !(p.variant & 2) && N >= nmin && N % 8 == 0 &&
                     !p.atomic && !partials && htiles >= 96 && !(p.variant & 2097152) &&
                     (c_dtype == UENC_F32 ? (epilogue == EPI_NONE || epilogue == EPI_RESIDUAL || epilogue == EPI_RELU) : true);
    const bool half_tile = mid || (big && !p.atomic && !partials && (K % BK3 == 0) && nt128_wins(M, N, K, epilogue, c_dtype, p.variant));
    if (prof) {
        // algorithmic HBM bytes: A, W and the output once, plus what the epilogue reads (residual / saved activation) or writes besides
        const double mn = (double)M * N;
        double bytes = (double)M * K * (a_dtype == UENC_F32 ? 4 : 2) + (double)N * K * 2 + mn * (c_dtype == UENC_F32 ? 4 : 2);
        if (epilogue == EPI_RESIDUAL) bytes += mn * 4;
        if (epilogue == EPI_MUL_DGELU || epilogue == EPI_MUL_DRELU) bytes += mn * 2;
        if (aux_out != nullptr) bytes += mn * 2;
        uenc_prof_begin(half_tile ? UENC_PROF_GEMM_NT128 : big ? UENC_PROF_GEMM_NT256 : UENC_PROF_GEMM_NT, 2.0 * batch * M * (double)N * K, stream, batch * bytes);
    }
    if (ln != nullptr && !(big || mid)) return UENC_EINVAL;       // (the small-shape kernels have no row-wide epilogue: the caller runs LayerNorm itself)
    // few output tiles (the decoder's M = 300-row GEMMs): the K-split 64 x 64 kernel
    const long tiles128 = (long)p.tiles_m * p.tiles_n;
    if (!big && batch == 1 && !p.atomic && !partials && tiles128 <= 32 && K >= 64 && !(p.variant & 1024)) {
        GemmNT q = p;
        q.tiles_m = (M + 63) / 64; q.tiles_n = (N + 63) / 64;
        const dim3 sgrid(q.tiles_m * q.tiles_n);
#define LAUNCHS(E, F)                                                                                                      \
        do {                                                                                                               \
            if (q.a_f32) hipLaunchKernelGGL((gemm_nt_skinny_kernel<E, F, 1>), sgrid, block, 0, stream, q);               \
            else hipLaunchKernelGGL((gemm_nt_skinny_kernel<E, F, 0>), sgrid, block, 0, stream, q);                       \
        } while (0)
        bool launched = true;
        if (c_dtype == UENC_F32) {
            if (epilogue == EPI_NONE) LAUNCHS(EPI_NONE, 1);
            else if (epilogue == EPI_RESIDUAL) LAUNCHS(EPI_RESIDUAL, 1);
            else if (epilogue == EPI_RELU) LAUNCHS(EPI_RELU, 1);
            else launched = false;
        } else {
            switch (epilogue) {
                case EPI_NONE: LAUNCHS(EPI_NONE, 0); break;
                case EPI_GELU: LAUNCHS(EPI_GELU, 0); break;
                case EPI_RELU: LAUNCHS(EPI_RELU, 0); break;
                case EPI_MUL_DGELU: LAUNCHS(EPI_MUL_DGELU, 0); break;
                case EPI_MUL_DRELU: LAUNCHS(EPI_MUL_DRELU, 0); break;
                default: launched = false; break;
            }
        }
#undef LAUNCHS
        if (!launched) return UENC_EINVAL;
        if (prof) uenc_prof_end(stream);
        UENC_LAUNCH_RET();
    }
    if (big || mid) {
        int rc = UENC_EINVAL;
        bool narrow = big && !half_tile && !p.atomic && !partials && nt192_wins(M, N, p.variant);
        if (ln != nullptr) {
            // fused LayerNorm: the row must sit in ONE column tile of an LDS-staged fp32 epilogue
            UENC_CHECK_ARG(epilogue == EPI_RESIDUAL && c_dtype == UENC_F32 && !p.atomic && !partials && N <= 256 && N % 8 == 0 && ldc == N &&
                           ln->gamma && ln->beta && (ln->y32 || ln->y16) && !(p.variant & 8192));
            UENC_CHECK_ARG((((uintptr_t)ln->gamma | (uintptr_t)ln->beta | (uintptr_t)ln->y32 | (uintptr_t)ln->y16) & 15) == 0 && ((uintptr_t)ln->stats & 7) == 0);
            narrow = narrow && N <= 192;
            p.ln_gamma = ln->gamma; p.ln_beta = ln->beta; p.ln_y32 = ln->y32; p.ln_y16 = (bf16*)ln->y16; p.ln_stats = (float2*)ln->stats; p.ln_eps = ln->eps;
        }
#define LAUNCH2(E, F) rc = half_tile ? launch_nt128<E, F>(p, stream) : (p.variant & 128) ? launch_nt256<E, F, 0>(p, stream) : narrow ? launch_nt256<E, F, 1, 192>(p, stream) : launch_nt256<E, F, 1>(p, stream)   /* bit 128: the two-stage loop, for A/B */
        if (ln != nullptr) {          // (checked above: fp32 residual epilogue, one column tile)
            rc = half_tile ? launch_nt128<EPI_RESIDUAL, 1, true>(p, stream)
                           : narrow ? launch_nt256<EPI_RESIDUAL, 1, 1, 192, true>(p, stream) : launch_nt256<EPI_RESIDUAL, 1, 1, BN2, true>(p, stream);
        } else if (c_dtype == UENC_F32) {
            if (epilogue == EPI_NONE) LAUNCH2(EPI_NONE, 1);
            else if (epilogue == EPI_RESIDUAL) LAUNCH2(EPI_RESIDUAL, 1);
            else if (epilogue == EPI_RELU) LAUNCH2(EPI_RELU, 1);
        } else {
            switch (epilogue) {
                case EPI_NONE: LAUNCH2(EPI_NONE, 0); break;
                case EPI_GELU: LAUNCH2(EPI_GELU, 0); break;
                case EPI_RELU: LAUNCH2(EPI_RELU, 0); break;
                case EPI_MUL_DGELU: LAUNCH2(EPI_MUL_DGELU, 0); break;
                case EPI_MUL_DRELU: LAUNCH2(EPI_MUL_DRELU, 0); break;
                case EPI_RESIDUAL: LAUNCH2(EPI_RESIDUAL, 0); break;       // fp32 residual added, bf16 result (a sum that only feeds the next GEMM)
                default: break;
            }
        }
#undef LAUNCH2
        if (rc != UENC_OK) return rc;
        if (prof) uenc_prof_end(stream);
        UENC_LAUNCH_RET();
    }
#define LAUNCH(E, F) hipLaunchKernelGGL((gemm_nt_kernel<E, F>), grid, block, 0, stream, p)
    if (c_dtype == UENC_F32) {
        if (epilogue == EPI_NONE) LAUNCH(EPI_NONE, 1);
        else if (epilogue == EPI_RESIDUAL) LAUNCH(EPI_RESIDUAL, 1);
        else if (epilogue == EPI_RELU) LAUNCH(EPI_RELU, 1);
        else return UENC_EINVAL;
    } else {
        switch (epilogue) {
            case EPI_NONE: LAUNCH(EPI_NONE, 0); break;
            case EPI_GELU: LAUNCH(EPI_GELU, 0); break;
            case EPI_RELU: LAUNCH(EPI_RELU, 0); break;
            case EPI_MUL_DGELU: LAUNCH(EPI_MUL_DGELU, 0); break;
            case EPI_MUL_DRELU: LAUNCH(EPI_MUL_DRELU, 0); break;
            default: return UENC_EINVAL;
        }
    }
#undef LAUNCH
    if (prof) uenc_prof_end(stream);
    UENC_LAUNCH_RET();
}

extern "C" int uenc_gemm_nt(const void* A, int a_dtype, long lda, const void* W, long ldw, void* C, int c_dtype, long ldc,
                            int M, int N, int K, const float* bias, int epilogue, const void* aux, long ldaux,
                            void* aux_out, long ldaux_out, float alpha, int splitk, int accumulate, hipStream_t stream) {
    return gemm_nt_impl(A, a_dtype, lda, W, ldw, C, c_dtype, ldc, M, N, K, bias, epilogue, aux, ldaux, aux_out, ldaux_out, alpha, splitk,
                        accumulate, 1, 0, 0, 0, stream);
}

// uenc_gemm_nt with alpha multiplied per SAMPLE: C[m][n] = epi(alpha * sample_scale[m / rows_per_sample] * (A W^T + bias)[m][n]).
// sample_scale: device pointer to ceil(M / rows_per_sample) floats.  Stochastic depth (timm DropPath, reference backbone/swin.py:279,
// 289) as an epilogue: a residual branch of image b is scaled by 0 or 1 / keep_prob while the GEMM runs over all images at once.
extern "C" int uenc_gemm_nt_scaled(const void* A, int a_dtype, long lda, const void* W, long ldw, void* C, int c_dtype, long ldc,
                                   int M, int N, int K, const float* bias, int epilogue, const void* aux, long ldaux,
                                   void* aux_out, long ldaux_out, float alpha, const float* sample_scale, int rows_per_sample,
                                   hipStream_t stream) {
    UENC_CHECK_ARG(sample_scale != nullptr && rows_per_sample > 0);
    return gemm_nt_impl(A, a_dtype, lda, W, ldw, C, c_dtype, ldc, M, N, K, bias, epilogue, aux, ldaux, aux_out, ldaux_out, alpha, 1,
                        0, 1, 0, 0, 0, stream, 0, sample_scale, rows_per_sample);
}

// y = LayerNorm(A W^T + bias + residual) with the LayerNorm inside the GEMM's epilogue (reference: Linear -> residual add -> nn.LayerNorm, e.g.
// pixel_decoder/msdeformattn.py:111-142, backbone/swin.py:262-295 at C = 192): C receives the pre-norm sum h (fp32, ldc == N; the LayerNorm
// backward reads it), y32 / y16 (either may be NULL) the normalised row as fp32 / bf16, stats (M, 2) = (mean, rstd) per row (may be NULL).
// Needs N <= 256, N % 8 == 0, bf16 A and a shape the LDS-staged kernels take; returns -1 (nothing launched) otherwise -- run
// uenc_gemm_nt + uenc_layernorm_fwd then.
extern "C" int uenc_gemm_nt_ln(const void* A, int a_dtype, long lda, const void* W, long ldw, void* C, long ldc, int M, int N, int K, const float* bias,
                               const void* residual, long ldres, const float* gamma, const float* beta, float eps, float* y32, void* y16, float* stats,
                               hipStream_t stream) {
    UENC_CHECK_ARG(residual != nullptr);
    NtLn ln = {gamma, beta, y32, y16, stats, eps};
    return gemm_nt_impl(A, a_dtype, lda, W, ldw, C, UENC_F32, ldc, M, N, K, bias, EPI_RESIDUAL, residual, ldres, nullptr, 0, 1.0f, 1, 0, 1, 0, 0, 0, stream, 0,
                        nullptr, 0, &ln);
}

// Split-K with stored partial sums: split s (of `splitk`) writes sum over its k-range to P + s * part_stride (fp32, ldp); the
// caller sums the `splitk` slices.  For long contractions with few output tiles (d(mask embeddings): 1536 x 256 outputs over
// 131072 pixels): fp32 atomics are per-lane 64-byte memory-side transactions, ~10 us per split here; stored tiles are ~1.
// `splitk` must be a fixed point of the k-range rounding (uenc_gemm_nt_splits gives it), so that every slice is written.
extern "C" int uenc_gemm_nt_splits(int K, int splitk) {
    const int kt = (K + BK - 1) / BK;
    if (splitk < 1) splitk = 1;
    if (splitk > kt) splitk = kt;
    const int klen = (kt + splitk - 1) / splitk;
    return (kt + klen - 1) / klen;
}

extern "C" int uenc_gemm_nt_partials(const void* A, int a_dtype, long lda, const void* W, long ldw, float* P, long ldp, long part_stride,
                                     int M, int N, int K, float alpha, int splitk, hipStream_t stream) {
    UENC_CHECK_ARG(splitk >= 1 && uenc_gemm_nt_splits(K, splitk) == splitk && part_stride >= (long)(M - 1) * ldp + N);
    return gemm_nt_impl(A, a_dtype, lda, W, ldw, P, UENC_F32, ldp, M, N, K, nullptr, EPI_NONE, nullptr, 0, nullptr, 0, alpha, splitk, 0, 1, 0, 0,
                        0, stream, part_stride);
}

// `batch` problems of one shape in one launch: problem b reads A + b * bsA, W + b * bsW and writes C + b * bsC (element
// strides); no bias / epilogue operands (EPI_NONE), fp32 or bf16 C, split-K allowed.  For skinny per-image GEMMs whose
// single launch cannot fill the chip (the mask-embedding gradient: 150 x 256 outputs over a 131072-long contraction).
extern "C" int uenc_gemm_nt_batched(const void* A, int a_dtype, long lda, long bsA, const void* W, long ldw, long bsW, void* C, int c_dtype,
                                    long ldc, long bsC, int batch, int M, int N, int K, float alpha, int splitk, int accumulate,
                                    hipStream_t stream) {
    UENC_CHECK_ARG(batch >= 1 && batch <= 65535);
    UENC_CHECK_ARG((bsA * (a_dtype == UENC_F32 ? 4 : 2)) % 16 == 0 && (bsW * 2) % 16 == 0 && (bsC * (c_dtype == UENC_F32 ? 4 : 2)) % 16 == 0);
    return gemm_nt_impl(A, a_dtype, lda, W, ldw, C, c_dtype, ldc, M, N, K, nullptr, EPI_NONE, nullptr, 0, nullptr, 0, alpha, splitk, accumulate,
                        batch, bsA, bsW, bsC, stream);
}

// ---------------------------------------------------------------------------------------------
// wgrad:  dW[n][k] += sum_m dY[m][n] * X[m][k]   (contraction over tokens), db[n] += sum_m dY[m][n]
// Both operands are read row-major [m][*] and transposed on the way into LDS: a thread loads an
// 8(m) x 8(col) bf16 block (eight 16-byte row pieces), transposes it in registers and writes eight
// 16-byte pieces, each 8 consecutive m of one column.  The LDS image is then the same [row][64 k]
// image gemm_nt uses, with m as the contraction index.  Waves 0-1 stage dY, waves 2-3 stage X.
// The fp32 tile is added to dW through LDS so that every atomic wave-instruction covers 256
// contiguous bytes (MI355X_MICROARCH.md "Global float atomics": the full-rate shape).
// ---------------------------------------------------------------------------------------------
struct GemmTN {
    const void* dY; int dy_f32; long ldy;
    const void* X; int x_f32; long ldx;
    float* dW; long ldw;
    float* db;
    int M, N, K;
    int tiles_n, tiles_k;
    int mlen;  // tokens per split (multiple of BK)
    int store; // tnbig only: 1 = the tile is stored (dW = ..., db = ...: single split, no prior zeroing), 0 = added atomically
    int variant;   // debug A/B switch (UENC_GEMM_VARIANT)
    float alpha;   // multiplies the sums (dW, db) before they are stored / added
};

__device__ __forceinline__ void transpose8x8(const u32x4 (&in)[8], u32x4 (&out)[8]) {
    // in[r] = row r (8 bf16 = 4 dwords), out[c] = column c as 8 bf16 (rows 0..7): one v_perm_b32 per output dword
#pragma unroll
    for (int c = 0; c < 8; ++c) {
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            const unsigned lo = in[2 * w][c >> 1], hi = in[2 * w + 1][c >> 1];
            out[c][w] = __builtin_amdgcn_perm(hi, lo, (c & 1) ? 0x07060302u : 0x05040100u);
        }
    }
}

__device__ __forceinline__ void tn_small_body(const GemmTN& p, int tile, int msplit) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * (BM + BN) * BK * 2];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int wk = wave >> 1, wn = wave & 1;
    const int ntile = tile / p.tiles_k, ktile = tile - ntile * p.tiles_k;
    const int n0 = ntile * BN, k0 = ktile * BM;
    const int mbeg = msplit * p.mlen;
    const int mend = min(p.M, mbeg + p.mlen);
    const int nit = (mend - mbeg + BK - 1) / BK;

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // staging role: threads 0..127 -> dY (rows of the LDS image = n), 128..255 -> X (rows = k)
    const int role = t >> 7;            // wave-uniform
    const int ts = t & 127;
    const int mb = ts & 7, cb = ts >> 3;   // 8-row block along m, 8-col block along n / k
    const bool do_db = (p.db != nullptr) && (ktile == 0) && (role == 0);
    float bsum[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) bsum[c] = 0.f;

    u32x4 rin[8];
    auto gload = [&](int it) {
        const int mrow = mbeg + it * BK + mb * 8;
        const void* src = role == 0 ? p.dY : p.X;
        const long ld = role == 0 ? p.ldy : p.ldx;
        const int ncols = role == 0 ? p.N : p.K;
        const int col = (role == 0 ? n0 : k0) + cb * 8;
        if (role == 0 ? p.dy_f32 : p.x_f32) {
#pragma unroll
            for (int r = 0; r < 8; ++r) rin[r] = load_row8<true>(src, ld, mrow + r, mend, col, ncols, ncols);
        } else {
#pragma unroll
            for (int r = 0; r < 8; ++r) rin[r] = load_row8<false>(src, ld, mrow + r, mend, col, ncols, ncols);
        }
    };
    auto lstore = [&](int buf) {
        unsigned char* Ys = smem + buf * ((BM + BN) * BK * 2);
        unsigned char* dst = role == 0 ? Ys : Ys + BN * BK * 2;
        if (do_db) {
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                const bf16x8 v = *(const bf16x8*)&rin[r];
#pragma unroll
                for (int c = 0; c < 8; ++c) bsum[c] += (float)v[c];
            }
        }
        u32x4 ro[8];
        transpose8x8(rin, ro);
#pragma unroll
        for (int c = 0; c < 8; ++c) *(u32x4*)(dst + lds_off(cb * 8 + c, mb)) = ro[c];
    };

    gload(0);
    lstore(0);
    __syncthreads();
    const int fr = lane & 15, fg = lane >> 4;
    for (int it = 0; it < nit; ++it) {
        const int buf = it & 1;
        if (it + 1 < nit) gload(it + 1);
        const unsigned char* Ys = smem + buf * ((BM + BN) * BK * 2);
        const unsigned char* Xs = Ys + BN * BK * 2;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 kf[4], nf[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) kf[i] = *(const bf16x8*)(Xs + lds_off(wk * 64 + i * 16 + fr, ks * 4 + fg));
#pragma unroll
            for (int j = 0; j < 4; ++j) nf[j] = *(const bf16x8*)(Ys + lds_off(wn * 64 + j * 16 + fr, ks * 4 + fg));
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = mfma16(kf[i], nf[j], acc[i][j]);
        }
        if (it + 1 < nit) lstore(buf ^ 1);
        __syncthreads();
    }

    if (do_db) {
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            float v = bsum[c];
            v += __shfl_xor(v, 1); v += __shfl_xor(v, 2); v += __shfl_xor(v, 4);
            const int n = n0 + cb * 8 + c;
            if (mb == 0 && n < p.N) atomicAdd(p.db + n, v * p.alpha);
        }
    }

    // acc[i][j][r] = dW[n = n0 + wn*64 + j*16 + fr][k = k0 + wk*64 + i*16 + 4*fg + r]
    // two passes of 64 n-rows through a padded fp32 LDS tile [64][132]
    float* T = (float*)smem;
    const int LDT = 132;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        __syncthreads();
        if (wn == h) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    *(f32x4*)(T + (j * 16 + fr) * LDT + wk * 64 + i * 16 + 4 * fg) = acc[i][j];
        }
        __syncthreads();
        for (int row = wave; row < 64; row += 4) {
            const int n = n0 + h * 64 + row;
            if (n >= p.N) break;
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int k = k0 + lane + 64 * q;
                if (k < p.K) {
                    if (p.store) p.dW[(long)n * p.ldw + k] = T[row * LDT + lane + 64 * q] * p.alpha;
                    else atomicAdd(p.dW + (long)n * p.ldw + k, T[row * LDT + lane + 64 * q] * p.alpha);
                }
            }
        }
    }
}

__global__ __launch_bounds__(GEMM_THREADS) void gemm_tn_kernel(GemmTN p) {
    tn_small_body(p, blockIdx.x, blockIdx.y);
}

// Grouped form of the register-staged kernel (any M, fp32 or bf16 operands): the decoder's ~120 small weight gradients per
// step are each a 4-workgroup launch that lasts ~20 us of pure latency; queued and launched together they overlap.
struct TnSmallDesc {
    const void* dY; const void* X; float* dW; float* db;
    long ldy, ldx, ldw;
    int M, N, K, dy_f32, x_f32, tiles_k, mlen, nsplit, item_begin;
    float alpha;              // multiplies the sums; 0 is read as 1 (descriptors written before the field existed)
};

__global__ __launch_bounds__(GEMM_THREADS) void gemm_tn_grouped_small_kernel(const TnSmallDesc* __restrict__ table, int n) {
    const int item = blockIdx.x;
    int lo = 0, hi = n - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (table[mid].item_begin <= item) lo = mid; else hi = mid - 1;
    }
    const TnSmallDesc d = table[lo];
    GemmTN p;
    p.dY = d.dY; p.dy_f32 = d.dy_f32; p.ldy = d.ldy; p.X = d.X; p.x_f32 = d.x_f32; p.ldx = d.ldx;
    p.dW = d.dW; p.ldw = d.ldw; p.db = d.db; p.M = d.M; p.N = d.N; p.K = d.K;
    p.tiles_n = 0; p.tiles_k = d.tiles_k; p.mlen = d.mlen; p.store = 0; p.alpha = d.alpha == 0.f ? 1.f : d.alpha;
    const int local = item - d.item_begin;
    tn_small_body(p, local / d.nsplit, local % d.nsplit);
}

// table: n descriptors (device memory) of 96 bytes {dY, X, dW, db, ldy, ldx, ldw, M, N, K, dy_f32, x_f32, tiles_k, mlen, nsplit,
// item_begin, alpha (float; 0 = 1)}: row-major dY [M][N] / X [M][K], fp32 (ld % 4 == 0) or bf16 (ld % 8 == 0), 16-byte aligned, N % 8 == 0,
// K % 8 == 0; tiles_k = ceil(K / 128); mlen (tokens per split) % 64 == 0; descriptor i owns items
// [item_begin, item_begin + ceil(N / 128) * tiles_k * nsplit).  dW[n][k] += sum_m dY[m][n] X[m][k]; db[n] += sum_m dY[m][n].
extern "C" int uenc_gemm_tn_grouped_small(const void* table, int n, int total_items, double flops, hipStream_t stream) {
    UENC_CHECK_ARG(table && n > 0 && total_items > 0 && ((uintptr_t)table & 7) == 0);
    static_assert(sizeof(TnSmallDesc) == 96, "descriptor layout is part of the ABI");
    const bool prof = uenc_prof_on();
    if (prof) uenc_prof_begin(UENC_PROF_GEMM_TN, flops, stream);
    hipLaunchKernelGGL(gemm_tn_grouped_small_kernel, dim3((unsigned)total_items), dim3(GEMM_THREADS), 0, stream, (const TnSmallDesc*)table, n);
    if (prof) uenc_prof_end(stream);
    UENC_LAUNCH_RET();
}


// ---------------------------------------------------------------------------------------------
// Large-tile wgrad: dW[n][k] += sum_m dY[m][n] X[m][k] with a 256(k) x 256(n) output tile, 8 waves.
// Both operands are token-major, i.e. the contraction index m is the ROW of both images, so neither
// can be read as a row fragment.  They are brought in untransposed by LDS-DMA (64 tokens x 256
// columns x 2 operands = 64 KB per stage, two stages) and every MFMA fragment is fetched with the
// hardware transposing read ds_read_b64_tr_b16 (4 rows x 16 columns per 16-lane group): no register
// transposes, no staging VGPRs.  512-byte rows; 32-byte pieces are XOR-permuted by
// (row & 3) | ((row >> 3) & 1) << 2 on the DMA source side so that the 8 row-pieces a 32-lane half
// reads fall into 8 different 32-byte bank groups.  The token range is split over blockIdx.y; the
// fp32 tile is added to dW through LDS in 256-byte contiguous atomic bursts.
// ---------------------------------------------------------------------------------------------
template <int COLS>
__device__ __forceinline__ int tn_off(int row, int col) {        // byte offset of element (row, col) in a [64][COLS] bf16 image
    const int piece = (col >> 4) ^ ((row & 3) | (((row >> 3) & 1) << 2));      // 32-byte piece index, low 3 bits permuted
    return row * (COLS * 2) + piece * 32 + (col & 15) * 2;
}

// fragment with the contraction index on rows: lane (i = lane & 15 -> column c0 + i; k-slots j' = rows r0 + 8*(lane>>4) + j')
template <int COLS>
__device__ __forceinline__ bf16x8 tn_frag(const unsigned char* img, int r0, int c0, int lane) {
    const int fg = lane >> 4, i = lane & 15, q4 = i >> 2, p4 = i & 3;
    typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4;
    const int ra = r0 + 8 * fg + q4, col = c0 + 4 * p4;
    const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(img + tn_off<COLS>(ra, col)));
    const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(img + tn_off<COLS>(ra + 4, col)));
    bf16x8 r;
#pragma unroll
    for (int j = 0; j < 4; ++j) { r[j] = lo[j]; r[4 + j] = hi[j]; }
    return r;
}

// TILE = 256: 8 waves (2 x 4), wave tile 128(k) x 64(n), 128 KB LDS, one workgroup per CU.
// TILE = 128: 4 waves (2 x 2), wave tile  64(k) x 64(n),  64 KB LDS, two workgroups per CU (small N x K outputs, long token ranges).
template <int TILE>
__device__ __forceinline__ void tnbig_body(const GemmTN& p, int tile, int msplit) {
    constexpr int NTHR = TILE * 2, NWN = TILE / 64, WKT = (TILE == 256 ? 8 : 4);     // waves along n; 16-wide k tiles per wave
    constexpr int IMG = 64 * TILE * 2;                                                   // bytes of one operand image
    constexpr int RPI = NTHR * 16 / (TILE * 2);                                          // rows filled per DMA instruction
    constexpr int CPR = TILE / 8;                                                        // 16-byte slots per row
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];                 // 2 stages x (dY + X)
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int wk = wave / NWN, wn = wave % NWN;
    const int ntile = tile / p.tiles_k, ktile = tile - ntile * p.tiles_k;
    const int n0 = ntile * TILE, k0 = ktile * TILE;
    const int mbeg = msplit * p.mlen;
    const int mend = min(p.M, mbeg + p.mlen);
    const int nst = (mend - mbeg) / 64;

    f32x4 acc[WKT][4];
#pragma unroll
    for (int i = 0; i < WKT; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // DMA: instruction q fills LDS 16-byte slot L = q * NTHR + t of an image: row L / CPR, position P = L % CPR
    const bf16* ysrc[4];
    const bf16* xsrc[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int row = q * RPI + t / CPR, P = t % CPR;
        const int piece = (P >> 1) ^ ((row & 3) | (((row >> 3) & 1) << 2));
        const int col = piece * 16 + (P & 1) * 8;
        ysrc[q] = (const bf16*)p.dY + (long)(mbeg + row) * p.ldy + min(n0 + col, p.N - 8);
        xsrc[q] = (const bf16*)p.X + (long)(mbeg + row) * p.ldx + min(k0 + col, p.K - 8);
    }
    typedef __attribute__((address_space(3))) void lds_void;
    auto issue = [&](int st, int stage) {
        unsigned char* Ys = smem + stage * (2 * IMG);
        unsigned char* Xs = Ys + IMG;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            __builtin_amdgcn_global_load_lds(ysrc[q] + (long)st * 64 * p.ldy, (lds_void*)(Ys + (q * NTHR + wave * 64) * 16), 16, 0, 0);
            __builtin_amdgcn_global_load_lds(xsrc[q] + (long)st * 64 * p.ldx, (lds_void*)(Xs + (q * NTHR + wave * 64) * 16), 16, 0, 0);
        }
    };

    // bias gradient (k-tile 0 only): thread owns 8 columns (one 16-byte chunk) and 4 of the 64 rows of each stage
    const bool do_db = (p.db != nullptr) && (ktile == 0);
    float bsum[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};

    if (nst > 0) issue(0, 0);
    for (int st = 0; st < nst; ++st) {
        __syncthreads();
        if (st + 1 < nst) issue(st + 1, (st + 1) & 1);
        const unsigned char* Ys = smem + (st & 1) * (2 * IMG);
        const unsigned char* Xs = Ys + IMG;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 kf[WKT], nf[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) nf[j] = tn_frag<TILE>(Ys, ks * 32, wn * 64 + j * 16, lane);
#pragma unroll
            for (int i = 0; i < WKT; ++i) kf[i] = tn_frag<TILE>(Xs, ks * 32, wk * (WKT * 16) + i * 16, lane);
#pragma unroll
            for (int i = 0; i < WKT; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = mfma16(kf[i], nf[j], acc[i][j]);
        }
        if (do_db) {
            const int c8 = (t % CPR) * 8, rg = t / CPR;       // 16 row groups
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
                const bf16x8 v = *(const bf16x8*)(Ys + tn_off<TILE>(rg + 16 * rr, c8));
#pragma unroll
                for (int c = 0; c < 8; ++c) bsum[c] += (float)v[c];
            }
        }
    }

    float* T = (float*)smem;
    constexpr int LDT = TILE + 4;
    if (do_db) {            // reduce the 16 row-groups through LDS, then one atomic per column
        __syncthreads();
        const int c8 = (t % CPR) * 8, rg = t / CPR;
#pragma unroll
        for (int c = 0; c < 8; ++c) T[rg * TILE + c8 + c] = bsum[c];
        __syncthreads();
        if (t < TILE) {
            float v = 0.f;
#pragma unroll
            for (int g = 0; g < 16; ++g) v += T[g * TILE + t];
            if (n0 + t < p.N) { if (p.store == 1) p.db[n0 + t] = v * p.alpha; else atomicAdd(p.db + n0 + t, v * p.alpha); }
        }
    }
    // acc[i][j][r] = dW[n = n0 + wn*64 + j*16 + fr][k = k0 + wk*WKT*16 + i*16 + 4*fg + r]; passes of 64 n-rows
    const int fr = lane & 15, fg = lane >> 4;
#pragma unroll
    for (int ps = 0; ps < NWN; ++ps) {
        __syncthreads();
        if (wn == ps) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int i = 0; i < WKT; ++i) *(f32x4*)(T + (j * 16 + fr) * LDT + wk * (WKT * 16) + i * 16 + 4 * fg) = acc[i][j];
        }
        __syncthreads();
        for (int row = wave; row < 64; row += NTHR / 64) {
            const int n = n0 + ps * 64 + row;
            if (n >= p.N) break;
#pragma unroll
            for (int q = 0; q < TILE / 64; ++q) {
                const int k = k0 + lane + 64 * q;
                if (k < p.K) {
                    if (p.store) p.dW[(long)n * p.ldw + k] = T[row * LDT + lane + 64 * q] * p.alpha;
                    else atomicAdd(p.dW + (long)n * p.ldw + k, T[row * LDT + lane + 64 * q] * p.alpha);
                }
            }
        }
    }
}

// Deep-prefetch form of the same tile (the default): one 64-token stage ahead of the MFMAs (above) leaves 0.86 us -- the stage's
// 8.4 MFLOP at the CU's peak -- for a DMA that takes 1-2.5 us under load, so the loop ran at the memory LATENCY (~0.7 PFLOP/s
// whatever the item order or the cache level the panels came from).  Here the ring holds four 32-token stages (one MFMA k-step
// each, same 128 / 64 KB), the DMA stream runs three stages = 96 tokens ahead, waits are counted (`vmcnt(8)` leaves two stages
// in flight) and the barrier is a raw s_barrier that does not drain the DMA queue: a stage is read only after the issuing
// waves' counted wait AND the barrier every wave passes after it; its slot is refilled one barrier after its last read.
template <int TILE>
__device__ __forceinline__ void tnbig_body_deep(const GemmTN& p, int tile, int msplit) {
    constexpr int NTHR = TILE * 2, NWN = TILE / 64, WKT = (TILE == 256 ? 8 : 4);
    constexpr int SR = 32, NS = 4, D = NS - 1;                                           // rows per stage, ring slots, stages in flight
    constexpr int IMG = SR * TILE * 2;                                                   // bytes of one operand image
    constexpr int RPI = NTHR * 16 / (TILE * 2);                                          // rows filled per DMA instruction (16)
    constexpr int CPR = TILE / 8;                                                        // 16-byte slots per row
    static_assert(RPI == 16, "two DMA instructions per operand and stage");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];                 // NS stages x (dY + X)
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int wk = wave / NWN, wn = wave % NWN;
    const int ntile = tile / p.tiles_k, ktile = tile - ntile * p.tiles_k;
    const int n0 = ntile * TILE, k0 = ktile * TILE;
    const int mbeg = msplit * p.mlen;
    const int mend = min(p.M, mbeg + p.mlen);
    const int nst = (mend - mbeg) / SR;

    f32x4 acc[WKT][4];
#pragma unroll
    for (int i = 0; i < WKT; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const bf16* ysrc[2];
    const bf16* xsrc[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int row = q * RPI + t / CPR, P = t % CPR;
        const int piece = (P >> 1) ^ ((row & 3) | (((row >> 3) & 1) << 2));
        const int col = piece * 16 + (P & 1) * 8;
        ysrc[q] = (const bf16*)p.dY + (long)(mbeg + row) * p.ldy + min(n0 + col, p.N - 8);
        xsrc[q] = (const bf16*)p.X + (long)(mbeg + row) * p.ldx + min(k0 + col, p.K - 8);
    }
    typedef __attribute__((address_space(3))) void lds_void;
    auto issue = [&](int st) {
        unsigned char* Ys = smem + (st & (NS - 1)) * (2 * IMG);
        unsigned char* Xs = Ys + IMG;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            __builtin_amdgcn_global_load_lds(ysrc[q] + (long)st * SR * p.ldy, (lds_void*)(Ys + (q * NTHR + wave * 64) * 16), 16, 0, 0);
            __builtin_amdgcn_global_load_lds(xsrc[q] + (long)st * SR * p.ldx, (lds_void*)(Xs + (q * NTHR + wave * 64) * 16), 16, 0, 0);
        }
    };

    const bool do_db = (p.db != nullptr) && (ktile == 0);
    float bsum[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};

#pragma unroll
    for (int i = 0; i < D; ++i)
        if (i < nst) issue(i);
    for (int st = 0; st < nst; ++st) {
        // stage st has landed (this wave's pieces) once at most the min(D - 1, stages after st) younger stages are outstanding
        wait_vmcnt_upto8(4 * min(D - 1, nst - 1 - st));
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();                        // ... and everyone's; every wave is also past its reads of stage st - 1
        __builtin_amdgcn_sched_barrier(0);
        if (st + D < nst) issue(st + D);                     // into the slot of stage st - 1
        const unsigned char* Ys = smem + (st & (NS - 1)) * (2 * IMG);
        const unsigned char* Xs = Ys + IMG;
        bf16x8 kf[WKT], nf[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) nf[j] = tn_frag<TILE>(Ys, 0, wn * 64 + j * 16, lane);
#pragma unroll
        for (int i = 0; i < WKT; ++i) kf[i] = tn_frag<TILE>(Xs, 0, wk * (WKT * 16) + i * 16, lane);
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int i = 0; i < WKT; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = mfma16(kf[i], nf[j], acc[i][j]);
        __builtin_amdgcn_s_setprio(0);
        if (do_db) {
            const int c8 = (t % CPR) * 8, rg = t / CPR;       // 16 row groups
#pragma unroll
            for (int rr = 0; rr < 2; ++rr) {
                const bf16x8 v = *(const bf16x8*)(Ys + tn_off<TILE>(rg + 16 * rr, c8));
#pragma unroll
                for (int c = 0; c < 8; ++c) bsum[c] += (float)v[c];
            }
        }
    }

    float* T = (float*)smem;
    constexpr int LDT = TILE + 4;
    if (do_db) {            // reduce the 16 row-groups through LDS, then one atomic per column
        __syncthreads();
        const int c8 = (t % CPR) * 8, rg = t / CPR;
#pragma unroll
        for (int c = 0; c < 8; ++c) T[rg * TILE + c8 + c] = bsum[c];
        __syncthreads();
        if (t < TILE) {
            float v = 0.f;
#pragma unroll
            for (int g = 0; g < 16; ++g) v += T[g * TILE + t];
            if (n0 + t < p.N) { if (p.store == 1) p.db[n0 + t] = v * p.alpha; else atomicAdd(p.db + n0 + t, v * p.alpha); }
        }
    }
    const int fr = lane & 15, fg = lane >> 4;
#pragma unroll
    for (int ps = 0; ps < NWN; ++ps) {
        __syncthreads();
        if (wn == ps) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int i = 0; i < WKT; ++i) *(f32x4*)(T + (j * 16 + fr) * LDT + wk * (WKT * 16) + i * 16 + 4 * fg) = acc[i][j];
        }
        __syncthreads();
        for (int row = wave; row < 64; row += NTHR / 64) {
            const int n = n0 + ps * 64 + row;
            if (n >= p.N) break;
#pragma unroll
            for (int q = 0; q < TILE / 64; ++q) {
                const int k = k0 + lane + 64 * q;
                if (k < p.K) {
                    if (p.store) p.dW[(long)n * p.ldw + k] = T[row * LDT + lane + 64 * q] * p.alpha;
                    else atomicAdd(p.dW + (long)n * p.ldw + k, T[row * LDT + lane + 64 * q] * p.alpha);
                }
            }
        }
    }
}

// Hand-scheduled form of the deep-prefetch loop.  In tnbig_body_deep hipcc (ROCm 7.2) puts `s_waitcnt vmcnt(0)` in front of the
// first transposing LDS read of every stage (an LDS read with a memory operand waits for ALL LDS-DMA in flight) and
// `lgkmcnt(0)` in front of the first MFMA: the DMA ring is drained every 32 tokens and no read overlaps an MFMA -- it measured
// 10-30 % SLOWER than the two-stage loop.  Here the fragment reads are inline asm (invisible to the waitcnt pass), issued at
// most 14 deep (lgkmcnt is a 4-bit counter), each group of four MFMAs waits with a counted lgkmcnt for exactly the fragment it
// consumes, and the bias gradient comes from four extra MFMAs against an all-ones fragment instead of LDS row sums.
template <int ROW4>
__device__ __forceinline__ void ds_tr_pair(bf16x8& f, unsigned addr) {
    bf16x4 lo, hi;
    asm volatile("ds_read_b64_tr_b16 %0, %2\n\tds_read_b64_tr_b16 %1, %2 offset:%3" : "=&v"(lo), "=&v"(hi) : "v"(addr), "n"(ROW4) : "memory");
#pragma unroll
    for (int j = 0; j < 4; ++j) { f[j] = lo[j]; f[4 + j] = hi[j]; }
}

template <int TILE>
__device__ __forceinline__ void tnbig_body_asm(const GemmTN& p, int tile, int msplit) {
    constexpr int NTHR = TILE * 2, NWN = TILE / 64, WKT = (TILE == 256 ? 8 : 4);
    constexpr int SR = 32, NS = 4, D = NS - 1;
    constexpr int IMG = SR * TILE * 2;
    constexpr int RPI = NTHR * 16 / (TILE * 2);
    constexpr int CPR = TILE / 8;
    constexpr int ROW4 = 4 * TILE * 2;                                                   // byte distance of rows r and r + 4
    static_assert(RPI == 16, "two DMA instructions per operand and stage");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int wk = wave / NWN, wn = wave % NWN;
    const int ntile = tile / p.tiles_k, ktile = tile - ntile * p.tiles_k;
    const int n0 = ntile * TILE, k0 = ktile * TILE;
    const int mbeg = msplit * p.mlen;
    const int mend = min(p.M, mbeg + p.mlen);
    const int nst = (mend - mbeg) / SR;

    f32x4 acc[WKT][4];
#pragma unroll
    for (int i = 0; i < WKT; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    f32x4 accb[4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};

    const bf16* ysrc[2];
    const bf16* xsrc[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int row = q * RPI + t / CPR, P = t % CPR;
        const int piece = (P >> 1) ^ ((row & 3) | (((row >> 3) & 1) << 2));
        const int col = piece * 16 + (P & 1) * 8;
        ysrc[q] = (const bf16*)p.dY + (long)(mbeg + row) * p.ldy + min(n0 + col, p.N - 8);
        xsrc[q] = (const bf16*)p.X + (long)(mbeg + row) * p.ldx + min(k0 + col, p.K - 8);
    }
    typedef __attribute__((address_space(3))) void lds_void;
    auto issue = [&](int st) {
        unsigned char* Ys = smem + (st & (NS - 1)) * (2 * IMG);
        unsigned char* Xs = Ys + IMG;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            __builtin_amdgcn_global_load_lds(ysrc[q] + (long)st * SR * p.ldy, (lds_void*)(Ys + (q * NTHR + wave * 64) * 16), 16, 0, 0);
            __builtin_amdgcn_global_load_lds(xsrc[q] + (long)st * SR * p.ldx, (lds_void*)(Xs + (q * NTHR + wave * 64) * 16), 16, 0, 0);
        }
    };
    // LDS byte addresses (relative to a stage's dY image) of this lane's first read of every fragment; the second read is ROW4 further
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem;
    unsigned anf[4], akf[WKT];
    {
        const int fg = lane >> 4, i16 = lane & 15, q4 = i16 >> 2, p4 = i16 & 3;
        const int ra = 8 * fg + q4;
#pragma unroll
        for (int j = 0; j < 4; ++j) anf[j] = lds0 + tn_off<TILE>(ra, wn * 64 + j * 16 + 4 * p4);
#pragma unroll
        for (int i = 0; i < WKT; ++i) akf[i] = lds0 + IMG + tn_off<TILE>(ra, wk * (WKT * 16) + i * 16 + 4 * p4);
    }
    const bool do_db = (p.db != nullptr) && (ktile == 0) && (wk == 0);
    bf16x8 ones;
#pragma unroll
    for (int j = 0; j < 8; ++j) ones[j] = (bf16)1.0f;

#pragma unroll
    for (int i = 0; i < D; ++i)
        if (i < nst) issue(i);
    for (int st = 0; st < nst; ++st) {
        wait_vmcnt_upto8(4 * min(D - 1, nst - 1 - st));
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        if (st + D < nst) issue(st + D);
        const unsigned so = (unsigned)((st & (NS - 1)) * (2 * IMG));
        bf16x8 kf[WKT], nf[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) ds_tr_pair<ROW4>(nf[j], anf[j] + so);
#pragma unroll
        for (int i = 0; i < 3; ++i) ds_tr_pair<ROW4>(kf[i], akf[i] + so);
#pragma unroll
        for (int i = 0; i < WKT; ++i) {
            // reads issued so far: nf + kf[0 .. min(i + 2, WKT - 1)]; kf[i] is complete once at most 2 * (min(i + 2, WKT - 1) - i) pairs remain
            if (i + 2 < WKT) asm volatile("s_waitcnt lgkmcnt(4)" ::: "memory");
            else if (i + 1 < WKT) asm volatile("s_waitcnt lgkmcnt(2)" ::: "memory");
            else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = mfma16(kf[i], nf[j], acc[i][j]);
            if (i + 3 < WKT) ds_tr_pair<ROW4>(kf[i + 3], akf[i + 3] + so);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (do_db) {
#pragma unroll
            for (int j = 0; j < 4; ++j) accb[j] = mfma16(ones, nf[j], accb[j]);
        }
    }

    float* T = (float*)smem;
    constexpr int LDT = TILE + 4;
    const int fr = lane & 15, fg = lane >> 4;
    if (do_db && fg == 0) {       // accb[j][r] = sum_m dY[m][n = wn * 64 + j * 16 + fr] for every r: the column sums, once per column
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int n = n0 + wn * 64 + j * 16 + fr;
            if (n < p.N) { if (p.store == 1) p.db[n] = accb[j][0] * p.alpha; else atomicAdd(p.db + n, accb[j][0] * p.alpha); }
        }
    }
#pragma unroll
    for (int ps = 0; ps < NWN; ++ps) {
        __syncthreads();
        if (wn == ps) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int i = 0; i < WKT; ++i) *(f32x4*)(T + (j * 16 + fr) * LDT + wk * (WKT * 16) + i * 16 + 4 * fg) = acc[i][j];
        }
        __syncthreads();
        for (int row = wave; row < 64; row += NTHR / 64) {
            const int n = n0 + ps * 64 + row;
            if (n >= p.N) break;
#pragma unroll
            for (int q = 0; q < TILE / 64; ++q) {
                const int k = k0 + lane + 64 * q;
                if (k < p.K) {
                    if (p.store) p.dW[(long)n * p.ldw + k] = T[row * LDT + lane + 64 * q] * p.alpha;
                    else atomicAdd(p.dW + (long)n * p.ldw + k, T[row * LDT + lane + 64 * q] * p.alpha);
                }
            }
        }
    }
}

template <int TILE>
__global__ __launch_bounds__(TILE * 2) void gemm_tnbig_kernel(GemmTN p) {
    if (p.variant & 2048) tnbig_body<TILE>(p, blockIdx.x, blockIdx.y);      // bit 2048: the round-1 main loop, for A/B
    else if (p.variant & 4096) tnbig_body_deep<TILE>(p, blockIdx.x, blockIdx.y);   // bit 4096: deep ring, compiler-scheduled reads
    else tnbig_body_asm<TILE>(p, blockIdx.x, blockIdx.y);
}

// Grouped form: ONE launch computes many weight gradients.  The backward of a layer leaves 4 small-output wgrad GEMMs
// (e.g. 36 tiles of 256 x 256 for a 3072 x 768 weight) that cannot fill 256 CUs unless the token range is split ~7-way,
// each split paying a 256 KB atomic burst per tile and a pipeline fill.  Deferred and launched together (the operands
// stay alive in HBM: there are 288 GB of it), the GEMMs of several layers give thousands of full-length work items.
// Work item = (descriptor, output tile, token-range split); descriptors sit in device memory, item -> descriptor by
// binary search over the exclusive prefix sum `item_begin`.
struct TnGroupDesc {
    const void* dY; const void* X; float* dW; float* db;
    long ldy, ldx, ldw;
    int M, N, K, tiles_k;
    int mlen, nsplit, item_begin, store;
    float alpha; int pad_;    // alpha multiplies the sums; 0 is read as 1
};

// Item order (XCD_WINDOWS): what shares operand panels must run on ONE XCD at the same time, or every tile re-fetches its
// dY / X panels from the fabric (measured round 1: 44 GB per step against ~19 GB read once).  Logical items are ordered
// (descriptor, token split, tile) with the tile fastest, so W consecutive items -- W = the workgroups an XCD runs at once -- are
// the tiles of one weight over the SAME token range: together they read each 64-token panel piece once into the XCD's L2.
// Hardware block b runs on XCD slot b % 8 as that slot's (b / 8)-th block: window c of the logical list goes to slot c % 8.
// Placement is for speed only (any mapping gives the same sums).
template <int TILE, bool XCD_WINDOWS>
__global__ __launch_bounds__(TILE * 2) void gemm_tnbig_grouped_kernel(const TnGroupDesc* __restrict__ table, int n, int total, int deep) {
    int item = blockIdx.x;
    if (XCD_WINDOWS) {
        constexpr int WIN = TILE == 256 ? 32 : 64;
        const int x = blockIdx.x & 7, j = blockIdx.x >> 3;
        item = ((j / WIN) * 8 + x) * WIN + (j % WIN);
        if (item >= total) return;
    }
    int lo = 0, hi = n - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (table[mid].item_begin <= item) lo = mid; else hi = mid - 1;
    }
    const TnGroupDesc d = table[lo];
    GemmTN p;
    p.dY = d.dY; p.dy_f32 = 0; p.ldy = d.ldy; p.X = d.X; p.x_f32 = 0; p.ldx = d.ldx;
    p.dW = d.dW; p.ldw = d.ldw; p.db = d.db; p.M = d.M; p.N = d.N; p.K = d.K;
    p.tiles_n = 0; p.tiles_k = d.tiles_k; p.mlen = d.mlen; p.store = d.nsplit == 1 ? d.store : 0; p.variant = 0; p.alpha = d.alpha == 0.f ? 1.f : d.alpha;
    const int local = item - d.item_begin;
    int tile, msplit;
    if (XCD_WINDOWS) {
        const int ntiles = ((d.N + TILE - 1) / TILE) * d.tiles_k;
        tile = local % ntiles; msplit = local / ntiles;
    } else {
        tile = local / d.nsplit; msplit = local % d.nsplit;
    }
    if (deep == 2) tnbig_body_asm<TILE>(p, tile, msplit);
    else if (deep == 1) tnbig_body_deep<TILE>(p, tile, msplit);
    else tnbig_body<TILE>(p, tile, msplit);
}

// table: n descriptors (device memory) of 96 bytes {dY, X, dW, db, ldy, ldx, ldw, M, N, K, tiles_k, mlen, nsplit, item_begin, store, alpha (float; 0 = 1), 0}:
// bf16 row-major operands dY [M][N] / X [M][K] (16-byte aligned, ld % 8 == 0), M % 64 == 0, mlen % 64 == 0,
// tiles_k = ceil(K / tile); descriptor i owns items [item_begin, item_begin + ceil(N/tile) * tiles_k * nsplit).
// dW[n][k] += sum_m dY[m][n] X[m][k], db[n] += sum_m dY[m][n] (db may be NULL); store != 0 (needs nsplit == 1): "=" instead
// of "+=" with plain stores; store == 2: dW is stored, db still added (a bias gradient that also receives other contributions).  tile = 256 or 128.
extern "C" int uenc_gemm_tn_grouped(const void* table, int n, int total_items, int tile, double flops, hipStream_t stream) {
    UENC_CHECK_ARG(table && n > 0 && total_items > 0 && (tile == 256 || tile == 128) && ((uintptr_t)table & 7) == 0);
    static_assert(sizeof(TnGroupDesc) == 96, "descriptor layout is part of the ABI");
    static bool attr_set = false;
    if (!attr_set) {
        const void* fns[4] = {(const void*)gemm_tnbig_grouped_kernel<256, true>, (const void*)gemm_tnbig_grouped_kernel<256, false>,
                              (const void*)gemm_tnbig_grouped_kernel<128, true>, (const void*)gemm_tnbig_grouped_kernel<128, false>};
        for (int i = 0; i < 4; ++i) {
            hipError_t e = hipFuncSetAttribute(fns[i], hipFuncAttributeMaxDynamicSharedMemorySize, 4 * 64 * (i < 2 ? 256 : 128) * 2);
            if (e != hipSuccess) return (int)e;
        }
        attr_set = true;
    }
    const bool prof = uenc_prof_on();
    if (prof) uenc_prof_begin(UENC_PROF_GEMM_TN, flops, stream);          // flops: 2 * sum(M N K), for the profiler only
    const char* ev = getenv("UENC_GEMM_VARIANT");
    const bool windows = !(ev && (atoi(ev) & 512));                       // bit 512: the round-1 item order (tile-major, split fastest), for A/B
    const int evv = ev ? atoi(ev) : 0;
    const int deep = (evv & 2048) ? 0 : (evv & 4096) ? 1 : 2;             // bit 2048: the round-1 main loop; bit 4096: deep ring with compiler-scheduled reads
    const int win = 8 * (tile == 256 ? 32 : 64);
    const unsigned grid = windows ? (unsigned)((total_items + win - 1) / win * win) : (unsigned)total_items;
    if (tile == 256) {
        if (windows) hipLaunchKernelGGL((gemm_tnbig_grouped_kernel<256, true>), dim3(grid), dim3(512), 4 * 64 * 256 * 2, stream, (const TnGroupDesc*)table, n, total_items, deep);
        else hipLaunchKernelGGL((gemm_tnbig_grouped_kernel<256, false>), dim3(grid), dim3(512), 4 * 64 * 256 * 2, stream, (const TnGroupDesc*)table, n, total_items, deep);
    } else {
        if (windows) hipLaunchKernelGGL((gemm_tnbig_grouped_kernel<128, true>), dim3(grid), dim3(256), 4 * 64 * 128 * 2, stream, (const TnGroupDesc*)table, n, total_items, deep);
        else hipLaunchKernelGGL((gemm_tnbig_grouped_kernel<128, false>), dim3(grid), dim3(256), 4 * 64 * 128 * 2, stream, (const TnGroupDesc*)table, n, total_items, deep);
    }
    if (prof) uenc_prof_end(stream);
    UENC_LAUNCH_RET();
}

template <int TILE>
static int launch_tnbig(GemmTN& p, int M, int N, int K, int splitm, int max_split, hipStream_t stream) {
    p.tiles_n = (N + TILE - 1) / TILE; p.tiles_k = (K + TILE - 1) / TILE;
    const int tiles = p.tiles_n * p.tiles_k, stages = M / 64;
    const int per_cu = TILE == 256 ? 1 : 2;
    int sp = splitm > 0 ? splitm : (256 * per_cu + tiles - 1) / tiles;
    if (sp > max_split) sp = max_split;
    if (sp > stages / 8) sp = stages / 8 > 0 ? stages / 8 : 1;
    p.mlen = ((stages + sp - 1) / sp) * 64;
    sp = (M + p.mlen - 1) / p.mlen;
    constexpr int SHM = 4 * 64 * TILE * 2;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e2 = hipFuncSetAttribute((const void*)gemm_tnbig_kernel<TILE>, hipFuncAttributeMaxDynamicSharedMemorySize, SHM);
        if (e2 != hipSuccess) return (int)e2;
        attr_set = true;
    }
    const bool prof2 = uenc_prof_on();
    if (prof2) uenc_prof_begin(UENC_PROF_GEMM_TN, 2.0 * M * (double)N * K, stream);
    hipLaunchKernelGGL(gemm_tnbig_kernel<TILE>, dim3(tiles, sp), dim3(TILE * 2), SHM, stream, p);
    if (prof2) uenc_prof_end(stream);
    return UENC_OK;
}

static int gemm_tn_impl(const void* dY, int dy_dtype, long ldy, const void* X, int x_dtype, long ldx, float* dW, long ldw,
                        float* db, int M, int N, int K, int splitm, float alpha, hipStream_t stream) {
    UENC_CHECK_ARG(dY && X && dW && M > 0 && N > 0 && K > 0);
    UENC_CHECK_ARG(N % 8 == 0 && K % 8 == 0);
    UENC_CHECK_ARG(dy_dtype == UENC_F32 ? (ldy % 4 == 0) : (dy_dtype == UENC_BF16 && ldy % 8 == 0));
    UENC_CHECK_ARG(x_dtype == UENC_F32 ? (ldx % 4 == 0) : (x_dtype == UENC_BF16 && ldx % 8 == 0));
    UENC_CHECK_ARG(((uintptr_t)dY & 15) == 0 && ((uintptr_t)X & 15) == 0);
    GemmTN p;
    p.dY = dY; p.dy_f32 = (dy_dtype == UENC_F32); p.ldy = ldy; p.X = X; p.x_f32 = (x_dtype == UENC_F32); p.ldx = ldx;
    p.dW = dW; p.ldw = ldw; p.db = db; p.M = M; p.N = N; p.K = K; p.store = 0; p.alpha = alpha;
    { const char* e0 = getenv("UENC_GEMM_VARIANT"); p.variant = e0 ? atoi(e0) : 0; }
    // LDS-DMA paths: bf16 operands, whole 64-token stages.  256x256 tiles when the output has >= 20 of them (every split
    // of the token range costs a 256 KB atomic burst per tile, so splits are capped at 8); else 128x128 tiles, whose
    // outputs are small enough to split the token range much further.
    { const char* e = getenv("UENC_GEMM_VARIANT"); const int variant = e ? atoi(e) : 0;
      if (dy_dtype == UENC_BF16 && x_dtype == UENC_BF16 && M % 64 == 0 && M >= 2048 && !(variant & 4)) {
        int rc;
        if ((variant & 8) || (long)((N + 255) / 256) * ((K + 255) / 256) >= 20) rc = launch_tnbig<256>(p, M, N, K, splitm, 8, stream);
        else rc = launch_tnbig<128>(p, M, N, K, splitm, 64, stream);
        if (rc != UENC_OK) return rc;
        UENC_LAUNCH_RET();
      }
    }
    p.tiles_n = (N + BN - 1) / BN; p.tiles_k = (K + BM - 1) / BM;
    const int mt = (M + BK - 1) / BK;
    if (splitm < 1) {
        // enough workgroups to fill 256 CUs ~2x, but at least 8 k-steps per workgroup
        const int tiles = p.tiles_n * p.tiles_k;
        splitm = (512 + tiles - 1) / tiles;
        const int maxsplit = (mt + 7) / 8;
        if (splitm > maxsplit) splitm = maxsplit;
        if (splitm < 1) splitm = 1;
    }
    if (splitm > mt) splitm = mt;
    p.mlen = ((mt + splitm - 1) / splitm) * BK;
    splitm = (M + p.mlen - 1) / p.mlen;
    dim3 grid(p.tiles_n * p.tiles_k, splitm), block(GEMM_THREADS);
    const bool prof = uenc_prof_on();
    if (prof) uenc_prof_begin(UENC_PROF_GEMM_TN, 2.0 * M * (double)N * K, stream);
    hipLaunchKernelGGL(gemm_tn_kernel, grid, block, 0, stream, p);
    if (prof) uenc_prof_end(stream);
    UENC_LAUNCH_RET();
}

extern "C" int uenc_gemm_tn(const void* dY, int dy_dtype, long ldy, const void* X, int x_dtype, long ldx, float* dW, long ldw,
                            float* db, int M, int N, int K, int splitm, hipStream_t stream) {
    return gemm_tn_impl(dY, dy_dtype, ldy, X, x_dtype, ldx, dW, ldw, db, M, N, K, splitm, 1.0f, stream);
}

// dW += alpha * dY^T X, db += alpha * column sums of dY (the weight gradient of a branch whose output was scaled by alpha: DropPath)
extern "C" int uenc_gemm_tn_scaled(const void* dY, int dy_dtype, long ldy, const void* X, int x_dtype, long ldx, float* dW, long ldw,
                                   float* db, int M, int N, int K, int splitm, float alpha, hipStream_t stream) {
    return gemm_tn_impl(dY, dy_dtype, ldy, X, x_dtype, ldx, dW, ldw, db, M, N, K, splitm, alpha, stream);
}
