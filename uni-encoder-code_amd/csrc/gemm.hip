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
    float alpha;
};

__device__ __forceinline__ int lds_off(int row, int chunk) { return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4); }

// load 8 consecutive k of one row as bf16x8 (zero outside)
__device__ __forceinline__ u32x4 load_row8(const void* base, int is_f32, long ld, int row, int nrows, int k, int kend) {
    u32x4 z = {0u, 0u, 0u, 0u};
    if (row >= nrows || k >= kend) return z;
    if (is_f32) {
        const float* p = (const float*)base + (long)row * ld + k;
        float4 a = *(const float4*)p, b = *(const float4*)(p + 4);
        bf16x8 r = cvt8(a, b);
        return *(u32x4*)&r;
    }
    return *(const u32x4*)((const bf16*)base + (long)row * ld + k);
}

template <int EPI, int OUT_F32>
__global__ __launch_bounds__(GEMM_THREADS) void gemm_nt_kernel(GemmNT p) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * (BM + BN) * BK * 2];
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
        for (int i = 0; i < 4; ++i) {
            ra[i] = load_row8(p.A, p.a_f32, p.lda, m0 + srow + 32 * i, p.M, k, kend);
            rw[i] = load_row8(p.W, 0, p.ldw, n0 + srow + 32 * i, p.N, k, kend);
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

    // epilogue: lane holds C[m = .. + fr][n = .. + 4*fg + 0..3]
    const bool lead = (blockIdx.y == 0);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int m = m0 + wm * 64 + j * 16 + fr;
        if (m >= p.M) continue;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int n = n0 + wn * 64 + i * 16 + 4 * fg;
            if (n >= p.N) continue;
            float v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = acc[i][j][r];
            if (p.bias != nullptr && lead) {
                const float4 b = *(const float4*)(p.bias + n);
                v[0] += b.x; v[1] += b.y; v[2] += b.z; v[3] += b.w;
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] *= p.alpha;     // alpha * (A W^T + bias): per-sample DropPath scale
            if (EPI == EPI_GELU) {
                if (p.aux_out != nullptr) {
                    bf16x4 pre;
#pragma unroll
                    for (int r = 0; r < 4; ++r) pre[r] = (bf16)v[r];
                    *(bf16x4*)(p.aux_out + (long)m * p.ldaux_out + n) = pre;
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = gelu_f(v[r]);
            } else if (EPI == EPI_RELU) {
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
            } else if (EPI == EPI_RESIDUAL) {
                const float4 rr = *(const float4*)((const float*)p.aux + (long)m * p.ldaux + n);
                v[0] += rr.x; v[1] += rr.y; v[2] += rr.z; v[3] += rr.w;
            } else if (EPI == EPI_MUL_DGELU) {
                const bf16x4 pre = *(const bf16x4*)((const bf16*)p.aux + (long)m * p.ldaux + n);
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] *= dgelu_f((float)pre[r]);
            } else if (EPI == EPI_MUL_DRELU) {
                const bf16x4 post = *(const bf16x4*)((const bf16*)p.aux + (long)m * p.ldaux + n);
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = ((float)post[r] > 0.f) ? v[r] : 0.f;
            }
            if (OUT_F32) {
                float* c = (float*)p.C + (long)m * p.ldc + n;
                if (p.atomic) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) atomicAdd(c + r, v[r]);
                } else {
                    *(float4*)c = make_float4(v[0], v[1], v[2], v[3]);
                }
            } else {
                bf16x4 o;
#pragma unroll
                for (int r = 0; r < 4; ++r) o[r] = (bf16)v[r];
                *(bf16x4*)((bf16*)p.C + (long)m * p.ldc + n) = o;
            }
        }
    }
}

extern "C" int uenc_gemm_nt(const void* A, int a_dtype, long lda, const void* W, long ldw, void* C, int c_dtype, long ldc,
                            int M, int N, int K, const float* bias, int epilogue, const void* aux, long ldaux,
                            void* aux_out, long ldaux_out, float alpha, int splitk, int accumulate, hipStream_t stream) {
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
    p.alpha = alpha;
    dim3 grid(p.tiles_m * p.tiles_n, splitk), block(GEMM_THREADS);
    const bool prof = uenc_prof_on();
    if (prof) uenc_prof_begin(UENC_PROF_GEMM_NT, 2.0 * M * (double)N * K, stream);
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
};

__device__ __forceinline__ void transpose8x8(const u32x4 (&in)[8], u32x4 (&out)[8]) {
    // in[r] = row r (8 bf16 = 4 dwords), out[c] = column c as 8 bf16 (rows 0..7)
#pragma unroll
    for (int c = 0; c < 8; ++c) {
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            const unsigned lo = in[2 * w][c >> 1], hi = in[2 * w + 1][c >> 1];
            out[c][w] = (c & 1) ? ((lo >> 16) | (hi & 0xffff0000u)) : ((lo & 0xffffu) | (hi << 16));
        }
    }
}

__global__ __launch_bounds__(GEMM_THREADS) void gemm_tn_kernel(GemmTN p) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * (BM + BN) * BK * 2];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int wk = wave >> 1, wn = wave & 1;
    const int tile = blockIdx.x;
    const int ntile = tile / p.tiles_k, ktile = tile - ntile * p.tiles_k;
    const int n0 = ntile * BN, k0 = ktile * BM;
    const int mbeg = blockIdx.y * p.mlen;
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
        if (role == 0) {
            const int col = n0 + cb * 8;
#pragma unroll
            for (int r = 0; r < 8; ++r) rin[r] = load_row8(p.dY, p.dy_f32, p.ldy, mrow + r, mend, col, p.N);
        } else {
            const int col = k0 + cb * 8;
#pragma unroll
            for (int r = 0; r < 8; ++r) rin[r] = load_row8(p.X, p.x_f32, p.ldx, mrow + r, mend, col, p.K);
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
            if (mb == 0 && n < p.N) atomicAdd(p.db + n, v);
        }
    }

    // acc[i][j][r] = dW[n = n0 + wn*64 + j*16 + fr][k = k0 + wk*64 + i*16 + 4*fg + r]
    // two passes of 64 n-rows through a padded fp32 LDS tile [64][132]
    float* T = (float*)smem;
    const int LDT = 132;
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
                if (k < p.K) atomicAdd(p.dW + (long)n * p.ldw + k, T[row * LDT + lane + 64 * q]);
            }
        }
    }
}

extern "C" int uenc_gemm_tn(const void* dY, int dy_dtype, long ldy, const void* X, int x_dtype, long ldx, float* dW, long ldw,
                            float* db, int M, int N, int K, int splitm, hipStream_t stream) {
    UENC_CHECK_ARG(dY && X && dW && M > 0 && N > 0 && K > 0);
    UENC_CHECK_ARG(N % 8 == 0 && K % 8 == 0);
    UENC_CHECK_ARG(dy_dtype == UENC_F32 ? (ldy % 4 == 0) : (dy_dtype == UENC_BF16 && ldy % 8 == 0));
    UENC_CHECK_ARG(x_dtype == UENC_F32 ? (ldx % 4 == 0) : (x_dtype == UENC_BF16 && ldx % 8 == 0));
    UENC_CHECK_ARG(((uintptr_t)dY & 15) == 0 && ((uintptr_t)X & 15) == 0);
    GemmTN p;
    p.dY = dY; p.dy_f32 = (dy_dtype == UENC_F32); p.ldy = ldy; p.X = X; p.x_f32 = (x_dtype == UENC_F32); p.ldx = ldx;
    p.dW = dW; p.ldw = ldw; p.db = db; p.M = M; p.N = N; p.K = K;
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
