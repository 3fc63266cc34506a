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
__global__ __launch_bounds__(256) void upsample_bilinear_kernel(const float* __restrict__ in, float* __restrict__ out, long NC,
                                                                int Hi, int Wi, int Ho, int Wo, float sy, float sx) {
    const int wq = Wo >> 2;
    const long total = NC * (long)Ho * wq;
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const int xq = (int)(idx % wq);
        const long r = idx / wq;
        const int oy = (int)(r % Ho);
        const long nc = r / Ho;
        float fy = ((float)oy + 0.5f) * sy - 0.5f;
        fy = fy < 0.f ? 0.f : fy;
        const int y0 = min((int)fy, Hi - 1), y1 = min(y0 + 1, Hi - 1);
        const float ly = fy - (float)y0;
        const float* r0 = in + (nc * Hi + y0) * (long)Wi;
        const float* r1 = in + (nc * Hi + y1) * (long)Wi;
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
        *(float4*)(out + (nc * Ho + oy) * (long)Wo + xq * 4) = make_float4(o[0], o[1], o[2], o[3]);
    }
}

extern "C" int uenc_upsample_bilinear(const float* in, float* out, long NC, int Hi, int Wi, int Ho, int Wo, hipStream_t stream) {
    UENC_CHECK_ARG(in && out && NC > 0 && Hi > 0 && Wi > 0 && Ho > 0 && Wo > 0 && Wo % 4 == 0 && ((uintptr_t)out & 15) == 0);
    const long total = NC * (long)Ho * (Wo / 4);
    long blocks = (total + 255) / 256;
    if (blocks > 16384) blocks = 16384;
    hipLaunchKernelGGL(upsample_bilinear_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, in, out, NC, Hi, Wi, Ho, Wo,
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
struct CastDesc { const float* src; bf16* dst; int rows, cols, transpose, tiles_c; long tile_begin; };

__global__ __launch_bounds__(256) void cast_multi_kernel(const CastDesc* __restrict__ table, int n, long total_tiles) {
    __shared__ float tile[64][65];
    for (long tix = blockIdx.x; tix < total_tiles; tix += gridDim.x) {
        int lo = 0, hi = n - 1;
        while (lo < hi) {
            const int mid = (lo + hi + 1) >> 1;
            if (table[mid].tile_begin <= tix) lo = mid; else hi = mid - 1;
        }
        const CastDesc d = table[lo];
        const int local = (int)(tix - d.tile_begin);
        const int tr = local / d.tiles_c, tc = local - tr * d.tiles_c;
        const int r0 = tr * 64, c0 = tc * 64;
        const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
        if (!d.transpose) {
            for (int i = ty; i < 64; i += 4) {
                const int r = r0 + i, c = c0 + tx;
                if (r < d.rows && c < d.cols) d.dst[(long)r * d.cols + c] = (bf16)d.src[(long)r * d.cols + c];
            }
        } else {
            __syncthreads();
            for (int i = ty; i < 64; i += 4) {
                const int r = r0 + i, c = c0 + tx;
                tile[i][tx] = (r < d.rows && c < d.cols) ? d.src[(long)r * d.cols + c] : 0.f;
            }
            __syncthreads();
            for (int i = ty; i < 64; i += 4) {
                const int c = c0 + i, r = r0 + tx;
                if (c < d.cols && r < d.rows) d.dst[(long)c * d.rows + r] = (bf16)tile[tx][i];
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
