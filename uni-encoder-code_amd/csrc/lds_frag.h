// LDS images of [rows][32] bf16 matrices (one attention head slice) and the MFMA fragment reads on them.
#pragma once
#include "common.h"

// LDS image of a [rows][32] bf16 matrix: 64-byte rows, chunk c (16 B) of row r at c ^ ((r >> 1) & 3)
__device__ __forceinline__ int rm_off(int row, int chunk) { return row * 64 + ((chunk ^ ((row >> 1) & 3)) << 4); }

// fragment of 16 rows x 32 k: lane (row = r0 + (lane & 15), k = 8 * (lane >> 4) ..)
__device__ __forceinline__ bf16x8 frag_rows(const unsigned char* img, int r0, int fr, int fg) {
    return *(const bf16x8*)(img + rm_off(r0 + fr, fg));
}

// transposed fragment for an MFMA A operand A[i = d][k]: d = d0 + (lane & 15); k-slots 0..3 <- rows
// ra + 4*fg + 0..3, k-slots 4..7 <- rows rb + 4*fg + 0..3 (the order P^T / dS tiles come out of the
// previous MFMA's accumulators).  Two ds_read_b64_tr_b16.
__device__ __forceinline__ bf16x8 frag_tr(const unsigned char* img, int ra, int rb, int d0, int lane) {
    const int fg = lane >> 4, i = lane & 15, q4 = i >> 2, p4 = i & 3;
    const int chunk = (d0 >> 3) + (p4 >> 1), sub = (p4 & 1) << 3;
    const int row_a = ra + 4 * fg + q4, row_b = rb + 4 * fg + q4;
    typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4;
    const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(img + rm_off(row_a, chunk) + sub));
    const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(img + rm_off(row_b, chunk) + sub));
    bf16x8 r;
#pragma unroll
    for (int j = 0; j < 4; ++j) { r[j] = lo[j]; r[4 + j] = hi[j]; }
    return r;
}

