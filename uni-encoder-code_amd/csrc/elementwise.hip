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

extern "C" int uenc_version(void) { return 1; }
extern "C" const char* uenc_arch(void) { return "gfx950"; }
