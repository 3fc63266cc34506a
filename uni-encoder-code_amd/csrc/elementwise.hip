// Small HBM-bound helpers: weight casts (fp32 master -> bf16 operand, optionally transposed).
#include "common.h"

__global__ __launch_bounds__(256) void cast_f32_bf16_kernel(const float* __restrict__ src, bf16* __restrict__ dst, long n8) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n8; i += (long)gridDim.x * 256) {
        const float4 a = *(const float4*)(src + i * 8), b = *(const float4*)(src + i * 8 + 4);
        *(bf16x8*)(dst + i * 8) = cvt8(a, b);
    }
}

extern "C" int uenc_cast_f32_bf16(const float* src, void* dst, long n, hipStream_t stream) {
    UENC_CHECK_ARG(src && dst && n > 0 && n % 8 == 0);
    UENC_CHECK_ARG(((uintptr_t)src & 15) == 0 && ((uintptr_t)dst & 15) == 0);
    long blocks = (n / 8 + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(cast_f32_bf16_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, src, (bf16*)dst, n / 8);
    UENC_LAUNCH_RET();
}

// Inverted dropout on a bf16 tensor: out[i] = keep(i) ? in[i] / (1 - p) : 0, keep(i) = the index hash of common.h (attn_keep) -- a pure
// function of (seed, i), so the backward regenerates the mask instead of storing it (reference: nn.Dropout at
// pixel_decoder/msdeformattn.py:111-142, applied between this library's GEMM / LayerNorm kernels in training mode).  In place allowed.
__global__ __launch_bounds__(256) void dropout_bf16_kernel(const bf16* __restrict__ in, bf16* __restrict__ out, long n8, unsigned seed,
                                                           unsigned thresh, float inv_keep) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n8; i += (long)gridDim.x * 256) {
        const bf16x8 v = *(const bf16x8*)(in + i * 8);
        bf16x8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = attn_keep(seed, thresh, (unsigned long long)(i * 8 + j)) ? (bf16)((float)v[j] * inv_keep) : (bf16)0.f;
        *(bf16x8*)(out + i * 8) = o;
    }
}

extern "C" int uenc_dropout_bf16(const void* in, void* out, long n, unsigned seed, float p, hipStream_t stream) {
    UENC_CHECK_ARG(in && out && n > 0 && n % 8 == 0 && p >= 0.f && p < 1.f);
    UENC_CHECK_ARG((((uintptr_t)in | (uintptr_t)out) & 15) == 0);
    long blocks = (n / 8 + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(dropout_bf16_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, (const bf16*)in, (bf16*)out, n / 8, seed,
                       attn_drop_thresh(p), 1.0f / (1.0f - p));
    UENC_LAUNCH_RET();
}

// dst[c][r] = bf16(src[r][c]); 64x64 tiles through LDS (padded), src row-major [R][C].
__global__ __launch_bounds__(256) void cast_transpose_kernel(const float* __restrict__ src, bf16* __restrict__ dst, int R, int C) {
    __shared__ float tile[64][65];
    const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    for (int i = ty; i < 64; i += 4) {
        const int r = r0 + i, c = c0 + tx;
        tile[i][tx] = (r < R && c < C) ? src[(long)r * C + c] : 0.f;
    }
    __syncthreads();
    for (int i = ty; i < 64; i += 4) {
        const int c = c0 + i, r = r0 + tx;
        if (c < C && r < R) dst[(long)c * R + r] = (bf16)tile[tx][i];
    }
}

extern "C" int uenc_cast_transpose_f32_bf16(const float* src, void* dst, int R, int C, hipStream_t stream) {
    UENC_CHECK_ARG(src && dst && R > 0 && C > 0);
    dim3 grid((C + 63) / 64, (R + 63) / 64);
    hipLaunchKernelGGL(cast_transpose_kernel, grid, dim3(256), 0, stream, src, (bf16*)dst, R, C);
    UENC_LAUNCH_RET();
}

// Bilinear resize (align_corners = False) of (NC, Hi, Wi) fp32 planes to (NC, Ho, Wo): the final x4 mask upsample of
// reference model/oneformer_model.py:255-263 (2.5 GB of output at bs 2: pure HBM write).  Each thread produces four
// consecutive output pixels of a row (one 16-byte store); the <= 2 x 6 input values it needs come from L1 / L2.
// Grid (row quarters, Ho, planes): plane and output row come from the block index (the first version derived them from a flat 64-bit
// index -- two 64-bit divisions per thread, more VALU time than the store stream leaves room for).
__global__ __launch_bounds__(256) void upsample_bilinear_kernel(const float* __restrict__ in, float* __restrict__ out, int planes_per_z,
                                                                long NC, int Hi, int Wi, int Ho, int Wo, float sy, float sx) {
    const int oy = blockIdx.y;
    float fy = ((float)oy + 0.5f) * sy - 0.5f;
    fy = fy < 0.f ? 0.f : fy;
    const int y0 = min((int)fy, Hi - 1), y1 = min(y0 + 1, Hi - 1);
    const float ly = fy - (float)y0;
    const int wq = Wo >> 2;
    for (int pz = 0; pz < planes_per_z; ++pz) {
        const long nc = (long)blockIdx.z * planes_per_z + pz;
        if (nc >= NC) break;
        const float* r0 = in + (nc * Hi + y0) * (long)Wi;
        const float* r1 = in + (nc * Hi + y1) * (long)Wi;
        float* orow = out + (nc * Ho + oy) * (long)Wo;
        for (int xq = blockIdx.x * 256 + threadIdx.x; xq < wq; xq += gridDim.x * 256) {
            float o[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float fx = ((float)(xq * 4 + j) + 0.5f) * sx - 0.5f;
                fx = fx < 0.f ? 0.f : fx;
                const int x0 = min((int)fx, Wi - 1), x1 = min(x0 + 1, Wi - 1);
                const float lx = fx - (float)x0;
                const float top = r0[x0] + (r0[x1] - r0[x0]) * lx, bot = r1[x0] + (r1[x1] - r1[x0]) * lx;
                o[j] = top + (bot - top) * ly;
            }
            // streamed once, never re-read by this kernel: non-temporal stores keep the 2.5 GB out of the way of the input planes in L2
            f32x4 v = {o[0], o[1], o[2], o[3]};
            __builtin_nontemporal_store(v, (f32x4*)(orow + xq * 4));
        }
    }
}

extern "C" int uenc_upsample_bilinear(const float* in, float* out, long NC, int Hi, int Wi, int Ho, int Wo, hipStream_t stream) {
    UENC_CHECK_ARG(in && out && NC > 0 && Hi > 0 && Wi > 0 && Ho > 0 && Wo > 0 && Wo % 4 == 0 && ((uintptr_t)out & 15) == 0);
    UENC_CHECK_ARG(Ho <= 65535);
    const int wq = Wo / 4;
    const int ppz = (int)((NC + 65534) / 65535);
    const dim3 grid((unsigned)((wq + 255) / 256 > 8 ? 8 : (wq + 255) / 256), (unsigned)Ho, (unsigned)((NC + ppz - 1) / ppz));
    hipLaunchKernelGGL(upsample_bilinear_kernel, grid, dim3(256), 0, stream, in, out, ppz, NC, Hi, Wi, Ho, Wo,
                       (float)Hi / (float)Ho, (float)Wi / (float)Wo);
    UENC_LAUNCH_RET();
}

// Attention mask of the masked-attention decoder (reference oneformer_transformer_decoder.py:497-505): bilinear
// resize (align_corners = False) of the (Hi, Wi) mask logits of one (image, query) row to the (Ho, Wo) key map,
// blocked = sigmoid(v) < 0.5 <=> v < 0, and the "a fully blocked row attends everywhere" fix of :454 -- one
// workgroup per row, so the row-wide test is a block reduction instead of a second pass over HBM.
template <int VEC>   // VEC output pixels of a row per thread (4 when Wo % 4 == 0: one 32-bit store, else 1)
__global__ __launch_bounds__(1024) void attn_mask_kernel(const float* __restrict__ in, uint8_t* __restrict__ out, int Hi, int Wi,
                                                         int Ho, int Wo, float sy, float sx) {
    const float* src = in + (long)blockIdx.x * Hi * Wi;
    uint8_t* dst = out + (long)blockIdx.x * Ho * Wo;
    const int wq = Wo / VEC, total = Ho * wq;
    int open_any = 0;
    // Exact 2x / 4x reductions (the 1/8 and 1/16 key maps of a 1/4-resolution mask): the two taps of an output pixel are the source pixels
    // s x + s/2 - 1 and s x + s/2 with weights 1/2, so the 4 outputs of a thread read 8 (16) CONSECUTIVE floats of two source rows: two
    // (four) 16-byte loads per row, all in flight before the first use, instead of 16 dependent 4-byte loads per iteration.  The
    // arithmetic is the generic expression with hx = lx = hy = ly = 1/2, term for term (same bits).
    if (VEC == 4 && (Wi & 3) == 0 && ((Hi == 2 * Ho && Wi == 2 * Wo) || (Hi == 4 * Ho && Wi == 4 * Wo)) && (((uintptr_t)src) & 15) == 0) {
        const int S = Hi / Ho;
        for (int idx = threadIdx.x; idx < total; idx += 1024) {
            const int oy = idx / wq, xq = idx - oy * wq;
            const int y0 = S * oy + S / 2 - 1;
            const float4* r0 = (const float4*)(src + (long)y0 * Wi + (long)xq * 4 * S);
            const float4* r1 = (const float4*)(src + (long)(y0 + 1) * Wi + (long)xq * 4 * S);
            float a0[4], a1[4], b0[4], b1[4];        // taps x0 / x1 of rows y0 / y1 for the 4 outputs
            if (S == 2) {
                const float4 p0 = r0[0], p1 = r0[1], q0 = r1[0], q1 = r1[1];
                a0[0] = p0.x; a1[0] = p0.y; a0[1] = p0.z; a1[1] = p0.w; a0[2] = p1.x; a1[2] = p1.y; a0[3] = p1.z; a1[3] = p1.w;
                b0[0] = q0.x; b1[0] = q0.y; b0[1] = q0.z; b1[1] = q0.w; b0[2] = q1.x; b1[2] = q1.y; b0[3] = q1.z; b1[3] = q1.w;
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) { const float4 pv = r0[j], qv = r1[j]; a0[j] = pv.y; a1[j] = pv.z; b0[j] = qv.y; b1[j] = qv.z; }
            }
            uint32_t packed = 0;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float v = 0.5f * (0.5f * a0[j] + 0.5f * a1[j]) + 0.5f * (0.5f * b0[j] + 0.5f * b1[j]);
                packed |= (v < 0.f ? 1u : 0u) << (8 * j);
            }
            open_any |= packed != 0x01010101u;
            *(uint32_t*)(dst + (long)oy * Wo + xq * 4) = packed;
        }
    } else
    for (int idx = threadIdx.x; idx < total; idx += 1024) {
        const int oy = idx / wq, xq = idx - oy * wq;
        float fy = ((float)oy + 0.5f) * sy - 0.5f;
        fy = fy < 0.f ? 0.f : fy;
        const int y0 = min((int)fy, Hi - 1), y1 = min(y0 + 1, Hi - 1);
        const float ly = fy - (float)y0, hy = 1.f - ly;
        const float* r0 = src + (long)y0 * Wi;
        const float* r1 = src + (long)y1 * Wi;
        uint32_t packed = 0;
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
            float fx = ((float)(xq * VEC + j) + 0.5f) * sx - 0.5f;
            fx = fx < 0.f ? 0.f : fx;
            const int x0 = min((int)fx, Wi - 1), x1 = min(x0 + 1, Wi - 1);
            const float lx = fx - (float)x0, hx = 1.f - lx;
            const float v = hy * (hx * r0[x0] + lx * r0[x1]) + ly * (hx * r1[x0] + lx * r1[x1]);
            packed |= (v < 0.f ? 1u : 0u) << (8 * j);
        }
        if (VEC == 4) {
            open_any |= packed != 0x01010101u;
            *(uint32_t*)(dst + (long)oy * Wo + xq * 4) = packed;
        } else {
            open_any |= packed == 0u;
            dst[(long)oy * Wo + xq] = (uint8_t)packed;
        }
    }
    if (!__syncthreads_or(open_any))
        for (int idx = threadIdx.x; idx < Ho * Wo; idx += 1024) dst[idx] = 0;
}

extern "C" int uenc_attn_mask(const float* logits, uint8_t* mask, long rows, int Hi, int Wi, int Ho, int Wo, hipStream_t stream) {
    UENC_CHECK_ARG(logits && mask && rows > 0 && Hi > 0 && Wi > 0 && Ho > 0 && Wo > 0);
    const float sy = (float)Hi / (float)Ho, sx = (float)Wi / (float)Wo;
    if (Wo % 4 == 0 && ((uintptr_t)mask & 3) == 0)
        hipLaunchKernelGGL(attn_mask_kernel<4>, dim3((unsigned)rows), dim3(1024), 0, stream, logits, mask, Hi, Wi, Ho, Wo, sy, sx);
    else
        hipLaunchKernelGGL(attn_mask_kernel<1>, dim3((unsigned)rows), dim3(1024), 0, stream, logits, mask, Hi, Wi, Ho, Wo, sy, sx);
    UENC_LAUNCH_RET();
}

// Batched refresh of the bf16 operand copies of many fp32 master weights in ONE launch (plain or transposed), instead
// of ~650 tiny cast launches per training step.  `table` (device memory) holds one descriptor per tensor; a 64x64 tile
// index is mapped to its tensor by binary search over the exclusive prefix sum of tile counts.
// `transpose`: bit 0 = write the transpose; bits 4.. = leading dimension of dst in elements (0: dense -- cols, or rows when transposed), so that
// several sources can fill row / column blocks of ONE destination (two Linear weights stacked for a single GEMM).
struct CastDesc { const float* src; bf16* dst; int rows, cols, transpose, tiles_c; long tile_begin; };

__global__ __launch_bounds__(256) void cast_multi_kernel(const CastDesc* __restrict__ table, int n, long total_tiles) {
    __shared__ float tile[64][65];
    for (long tix = blockIdx.x; tix < total_tiles; tix += gridDim.x) {
        int lo = 0, hi = n - 1;
        while (lo < hi) {
            const int mid = (lo + hi + 1) >> 1;
            if (table[mid].tile_begin <= tix) lo = mid; else hi = mid - 1;
        }
        CastDesc d = table[lo];
        const long ldd = (d.transpose >> 4) ? (long)(d.transpose >> 4) : (long)((d.transpose & 1) ? d.rows : d.cols);
        d.transpose &= 1;
        const int local = (int)(tix - d.tile_begin);
        const int tr = local / d.tiles_c, tc = local - tr * d.tiles_c;
        const int r0 = tr * 64, c0 = tc * 64;
        const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
        if (!d.transpose && (d.cols & 3) == 0 && (ldd & 3) == 0 && ((((uintptr_t)d.src) & 15) | (((uintptr_t)d.dst) & 7)) == 0) {
            // 16 bytes in, 8 bytes out per lane: 16 lanes per 64-column row segment, 16 rows per pass
            const int cx = c0 + (threadIdx.x & 15) * 4;
            for (int i = threadIdx.x >> 4; i < 64; i += 16) {
                const int r = r0 + i;
                if (r < d.rows && cx < d.cols) {
                    const float4 v = *(const float4*)(d.src + (long)r * d.cols + cx);
                    bf16x4 o; o[0] = (bf16)v.x; o[1] = (bf16)v.y; o[2] = (bf16)v.z; o[3] = (bf16)v.w;
                    *(bf16x4*)(d.dst + (long)r * ldd + cx) = o;
                }
            }
        } else if (!d.transpose) {
            for (int i = ty; i < 64; i += 4) {
                const int r = r0 + i, c = c0 + tx;
                if (r < d.rows && c < d.cols) d.dst[(long)r * ldd + c] = (bf16)d.src[(long)r * d.cols + c];
            }
        } else {
            __syncthreads();
            for (int i = ty; i < 64; i += 4) {
                const int r = r0 + i, c = c0 + tx;
                tile[i][tx] = (r < d.rows && c < d.cols) ? d.src[(long)r * d.cols + c] : 0.f;
            }
            __syncthreads();
            if ((d.rows & 3) == 0 && (ldd & 3) == 0 && (((uintptr_t)d.dst) & 7) == 0) {       // 8 bytes out per lane: 16 lanes per 64-row output segment
                const int rx = (threadIdx.x & 15) * 4;
                for (int i = threadIdx.x >> 4; i < 64; i += 16) {
                    const int c = c0 + i, r = r0 + rx;
                    if (c < d.cols && r < d.rows) {
                        bf16x4 o;
#pragma unroll
                        for (int e = 0; e < 4; ++e) o[e] = (bf16)tile[rx + e][i];
                        *(bf16x4*)(d.dst + (long)c * ldd + r) = o;
                    }
                }
            } else {
                for (int i = ty; i < 64; i += 4) {
                    const int c = c0 + i, r = r0 + tx;
                    if (c < d.cols && r < d.rows) d.dst[(long)c * ldd + r] = (bf16)tile[tx][i];
                }
            }
        }
    }
}

// table: n descriptors of 40 bytes {src, dst, rows, cols, transpose, tiles_c, tile_begin} in device memory.
extern "C" int uenc_cast_multi(const void* table, int n, long total_tiles, hipStream_t stream) {
    UENC_CHECK_ARG(table && n > 0 && total_tiles > 0);
    long blocks = total_tiles < 8192 ? total_tiles : 8192;
    hipLaunchKernelGGL(cast_multi_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, (const CastDesc*)table, n, total_tiles);
    UENC_LAUNCH_RET();
}

extern "C" int uenc_version(void) { return 1; }
extern "C" const char* uenc_arch(void) { return "gfx950"; }


// ---- glue of the deformable encoder layer (pixel_decoder/ops/modules/ms_deform_attn.py:91-113) as three small kernels ----
// out16 = bf16(a + b), b repeating every `period` elements (period == n: same shape): the query = src + pos operand of
// the sampling-offset / attention-weight GEMM, without materialising the fp32 sum.
__global__ __launch_bounds__(256) void add_cast_kernel(const float* __restrict__ a, const float* __restrict__ b, bf16* __restrict__ out,
                                                       long n4, long period4) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        const float4 x = ((const float4*)a)[i], y = ((const float4*)b)[i % period4];
        bf16x4 o; o[0] = (bf16)(x.x + y.x); o[1] = (bf16)(x.y + y.y); o[2] = (bf16)(x.z + y.z); o[3] = (bf16)(x.w + y.w);
        ((bf16x4*)out)[i] = o;
    }
}

extern "C" int uenc_add_cast_bf16(const float* a, const float* b, void* out, long n, long period, hipStream_t stream) {
    UENC_CHECK_ARG(a && b && out && n > 0 && period > 0 && n % 4 == 0 && period % 4 == 0 && n % period == 0);
    UENC_CHECK_ARG((((uintptr_t)a | (uintptr_t)b) & 15) == 0 && ((uintptr_t)out & 7) == 0);
    long blocks = (n / 4 + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(add_cast_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, a, b, (bf16*)out, n / 4, period / 4);
    UENC_LAUNCH_RET();
}

// One thread per (row = image * Lq + query, head).  offaw: (rows, ld) fp32, columns [M][L][P][2] sampling offsets then
// [M][L*P] attention logits (the output of one GEMM over the concatenated weights).
//   loc[row][m][l][p][:] = ref[row or query][l][:] + off / (W_l, H_l)        (ms_deform_attn.py:103-106)
//   aw [row][m][:]       = softmax over the L*P logits                       (:101)
struct PrepP {
    const float* offaw; long ld;
    const float* ref; int ref_per_image;       // (N|1, Lq, L, 2)
    const int64_t* shapes;                     // (L, 2) H, W
    float* loc; float* aw;                     // forward outputs / backward: aw = saved softmax
    const float* dloc; const float* daw;       // backward inputs
    bf16* doffaw;                              // backward output (rows, ld) bf16
    long rows; int Lq, M, L, P;
};

// LPQ = L * P / 4 (> 0): every per-(row, head) vector is moved as 16-byte pieces held in registers; LPQ = 0: generic scalar form.
template <bool BWD, int LPQ>
__global__ __launch_bounds__(256) void msda_prep_kernel(PrepP p) {
    const long total = p.rows * p.M;
    const int LP = LPQ > 0 ? LPQ * 4 : p.L * p.P;
    constexpr int NV = LPQ > 0 ? LPQ * 4 : 16;
    for (long t = (long)blockIdx.x * 256 + threadIdx.x; t < total; t += (long)gridDim.x * 256) {
        const long row = t / p.M;
        const int m = (int)(t - row * p.M);
        const long o_off = row * p.ld + (long)m * LP * 2, o_lg = row * p.ld + (long)p.M * LP * 2 + (long)m * LP;
        float v[2 * NV], w[NV], a[NV];       // offsets or d(loc); logits or d(aw); softmax
        if (!BWD) {
            if (LPQ > 0) {
#pragma unroll
                for (int i = 0; i < 2 * LPQ; ++i) *(float4*)(v + 4 * i) = *(const float4*)(p.offaw + o_off + 4 * i);
#pragma unroll
                for (int i = 0; i < LPQ; ++i) *(float4*)(w + 4 * i) = *(const float4*)(p.offaw + o_lg + 4 * i);
            } else {
                for (int j = 0; j < NV; ++j) { w[j] = j < LP ? p.offaw[o_lg + j] : -3.0e38f; v[2 * j] = j < LP ? p.offaw[o_off + 2 * j] : 0.f; v[2 * j + 1] = j < LP ? p.offaw[o_off + 2 * j + 1] : 0.f; }
            }
            const long q = p.ref_per_image ? row : row % p.Lq;
            const float* rf = p.ref + q * p.L * 2;
            float mx = -3.0e38f;
#pragma unroll
            for (int j = 0; j < NV; ++j) if (j < LP) mx = fmaxf(mx, w[j]);
            float sum = 0.f;
#pragma unroll
            for (int j = 0; j < NV; ++j) { w[j] = j < LP ? __expf(w[j] - mx) : 0.f; sum += w[j]; }
            const float inv = 1.0f / sum;
#pragma unroll
            for (int j = 0; j < NV; ++j) {
                w[j] *= inv;
                if (j < LP) {
                    const int l = j / p.P;
                    v[2 * j] = rf[2 * l] + v[2 * j] / (float)p.shapes[2 * l + 1];
                    v[2 * j + 1] = rf[2 * l + 1] + v[2 * j + 1] / (float)p.shapes[2 * l];
                }
            }
            if (LPQ > 0) {
#pragma unroll
                for (int i = 0; i < 2 * LPQ; ++i) *(float4*)(p.loc + t * LP * 2 + 4 * i) = *(const float4*)(v + 4 * i);
#pragma unroll
                for (int i = 0; i < LPQ; ++i) *(float4*)(p.aw + t * LP + 4 * i) = *(const float4*)(w + 4 * i);
            } else {
                for (int j = 0; j < LP; ++j) { p.aw[t * LP + j] = w[j]; p.loc[t * LP * 2 + 2 * j] = v[2 * j]; p.loc[t * LP * 2 + 2 * j + 1] = v[2 * j + 1]; }
            }
        } else {
            if (LPQ > 0) {
#pragma unroll
                for (int i = 0; i < 2 * LPQ; ++i) *(float4*)(v + 4 * i) = *(const float4*)(p.dloc + t * LP * 2 + 4 * i);
#pragma unroll
                for (int i = 0; i < LPQ; ++i) { *(float4*)(w + 4 * i) = *(const float4*)(p.daw + t * LP + 4 * i); *(float4*)(a + 4 * i) = *(const float4*)(p.aw + t * LP + 4 * i); }
            } else {
                for (int j = 0; j < NV; ++j) {
                    w[j] = j < LP ? p.daw[t * LP + j] : 0.f; a[j] = j < LP ? p.aw[t * LP + j] : 0.f;
                    v[2 * j] = j < LP ? p.dloc[t * LP * 2 + 2 * j] : 0.f; v[2 * j + 1] = j < LP ? p.dloc[t * LP * 2 + 2 * j + 1] : 0.f;
                }
            }
            float dot = 0.f;
#pragma unroll
            for (int j = 0; j < NV; ++j) dot += a[j] * w[j];
            bf16 dof[2 * NV], dlg[NV];
#pragma unroll
            for (int j = 0; j < NV; ++j) {
                const int l = j < LP ? j / p.P : 0;
                dof[2 * j] = (bf16)(v[2 * j] / (float)p.shapes[2 * l + 1]);
                dof[2 * j + 1] = (bf16)(v[2 * j + 1] / (float)p.shapes[2 * l]);
                dlg[j] = (bf16)(a[j] * (w[j] - dot));
            }
            if (LPQ > 0) {
#pragma unroll
                for (int i = 0; i < LPQ; ++i) {
                    *(bf16x8*)(p.doffaw + o_off + 8 * i) = *(const bf16x8*)(dof + 8 * i);
                    *(bf16x4*)(p.doffaw + o_lg + 4 * i) = *(const bf16x4*)(dlg + 4 * i);
                }
            } else {
                for (int j = 0; j < LP; ++j) { p.doffaw[o_off + 2 * j] = dof[2 * j]; p.doffaw[o_off + 2 * j + 1] = dof[2 * j + 1]; p.doffaw[o_lg + j] = dlg[j]; }
            }
        }
    }
}

template <bool BWD>
static void prep_launch(const PrepP& p, hipStream_t stream) {
    long blocks = (p.rows * p.M + 255) / 256;
    if (blocks > 16384) blocks = 16384;
    const int LP = p.L * p.P;
    const bool vec = LP % 4 == 0 && p.ld % 8 == 0;       // 16-byte pieces stay aligned
    dim3 g((unsigned)blocks), b(256);
    if (vec && LP == 4) hipLaunchKernelGGL((msda_prep_kernel<BWD, 1>), g, b, 0, stream, p);
    else if (vec && LP == 8) hipLaunchKernelGGL((msda_prep_kernel<BWD, 2>), g, b, 0, stream, p);
    else if (vec && LP == 12) hipLaunchKernelGGL((msda_prep_kernel<BWD, 3>), g, b, 0, stream, p);
    else if (vec && LP == 16) hipLaunchKernelGGL((msda_prep_kernel<BWD, 4>), g, b, 0, stream, p);
    else hipLaunchKernelGGL((msda_prep_kernel<BWD, 0>), g, b, 0, stream, p);
}

static int prep_fill(PrepP& p, const float* ref, int ref_per_image, const int64_t* shapes, long rows, int Lq, int M, int L, int P, long ld) {
    if (!(ref && shapes && rows > 0 && Lq > 0 && M > 0 && L > 0 && P > 0 && L * P <= 16 && ld >= (long)M * L * P * 3)) return UENC_EINVAL;
    p.ref = ref; p.ref_per_image = ref_per_image; p.shapes = shapes; p.rows = rows; p.Lq = Lq; p.M = M; p.L = L; p.P = P; p.ld = ld;
    p.offaw = nullptr; p.loc = p.aw = nullptr; p.dloc = p.daw = nullptr; p.doffaw = nullptr;
    return UENC_OK;
}

extern "C" int uenc_msda_prep_fwd(const float* offaw, long ld, const float* ref, int ref_per_image, const int64_t* shapes, float* loc,
                                  float* aw, long rows, int Lq, int M, int L, int P, hipStream_t stream) {
    PrepP p;
    int rc = prep_fill(p, ref, ref_per_image, shapes, rows, Lq, M, L, P, ld);
    if (rc != UENC_OK) return rc;
    UENC_CHECK_ARG(offaw && loc && aw);
    UENC_CHECK_ARG((((uintptr_t)offaw | (uintptr_t)loc | (uintptr_t)aw) & 15) == 0);
    p.offaw = offaw; p.loc = loc; p.aw = aw;
    prep_launch<false>(p, stream);
    UENC_LAUNCH_RET();
}

// d(offaw) in bf16 (the dgrad / wgrad GEMM operand) from d(loc), d(softmaxed weights) and the saved softmax.
extern "C" int uenc_msda_prep_bwd(const float* dloc, const float* daw, const float* aw, const int64_t* shapes, void* doffaw, long ld,
                                  long rows, int Lq, int M, int L, int P, hipStream_t stream) {
    PrepP p;
    int rc = prep_fill(p, dloc, 0, shapes, rows, Lq, M, L, P, ld);
    if (rc != UENC_OK) return rc;
    UENC_CHECK_ARG(dloc && daw && aw && doffaw);
    UENC_CHECK_ARG((((uintptr_t)dloc | (uintptr_t)daw | (uintptr_t)aw | (uintptr_t)doffaw) & 15) == 0);
    p.dloc = dloc; p.daw = daw; p.aw = (float*)aw; p.doffaw = (bf16*)doffaw;
    prep_launch<true>(p, stream);
    UENC_LAUNCH_RET();
}

// out[seg][blk][c] = partial sums (blk < 128; the caller adds them) over rows [seg_start[seg], seg_start[seg + 1]) of every image of
// x16[row][c]: the per-level column sums that give the level-embedding gradient (level_embed enters the query through pos, msdeformattn.py:104-113).
__global__ __launch_bounds__(256) void segment_colsum_kernel(const bf16* __restrict__ x, long ld, int cols, const int64_t* __restrict__ seg_start,
                                                             int nseg, long rows_per_image, int images, float* __restrict__ out) {
    // thread = (8-column group cg, row lane r); 16-byte loads; the block's partial row is STORED (out[seg][block][cols]): hundreds
    // of workgroups adding into the same few hundred floats would run at the same-address atomic rate (~100 us here)
    __shared__ float red[256][8];
    const int seg = blockIdx.y;
    const long s0 = seg_start[seg], s1 = seg + 1 < nseg ? (long)seg_start[seg + 1] : rows_per_image;
    const long len = s1 - s0, n = len * images;
    const int ncg = cols >> 3, nr = 256 / ncg;
    const int cg = threadIdx.x % ncg, r = threadIdx.x / ncg;
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (r < nr)
        for (long i = (long)blockIdx.x * nr + r; i < n; i += (long)gridDim.x * nr) {
            const long img = i / len, row = s0 + (i - img * len);
            const bf16x8 v = *(const bf16x8*)(x + (img * rows_per_image + row) * ld + cg * 8);
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j] += (float)v[j];
        }
#pragma unroll
    for (int j = 0; j < 8; ++j) red[threadIdx.x][j] = r < nr ? acc[j] : 0.f;
    __syncthreads();
    if (r == 0) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float v = 0.f;
            for (int rr = 0; rr < nr; ++rr) v += red[cg + rr * ncg][j];
            out[((long)seg * gridDim.x + blockIdx.x) * cols + cg * 8 + j] = v;
        }
    }
}

extern "C" int uenc_segment_colsum(const void* x16, long ld, int cols, const int64_t* seg_start, int nseg, long rows_per_image, int images,
                                   float* out, hipStream_t stream) {
    UENC_CHECK_ARG(x16 && seg_start && out && cols > 0 && nseg > 0 && rows_per_image > 0 && images > 0 && ld >= cols);
    UENC_CHECK_ARG(cols % 8 == 0 && cols <= 2048 && ld % 8 == 0 && ((uintptr_t)x16 & 15) == 0);
    hipLaunchKernelGGL(segment_colsum_kernel, dim3(128, nseg), dim3(256), 0, stream, (const bf16*)x16, ld, cols, seg_start, nseg,
                       rows_per_image, images, out);
    UENC_LAUNCH_RET();
}
