// Shared device helpers for the gfx950 (CDNA4) kernels of libuenc_hip.so.
// Wave = 64 lanes; MFMA = v_mfma_f32_16x16x32_bf16 (fp32 accumulate).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __bf16 bf16;
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

#define UENC_OK 0
#define UENC_EINVAL (-1)

#define UENC_CHECK_ARG(cond) \
    do {                     \
        if (!(cond)) return UENC_EINVAL; \
    } while (0)

#define UENC_LAUNCH_RET()                       \
    do {                                        \
        hipError_t e__ = hipGetLastError();     \
        return e__ == hipSuccess ? UENC_OK : (int)e__; \
    } while (0)

// The C ABI itself (dtype tags, epilogue codes, every entry point): each definition in csrc/*.hip is compiled against its
// declaration, so hipcc rejects any drift between include/uenc.h and the library.
#define UENC_STREAM_T hipStream_t
#include "../../include/uenc.h"

// D = A(16x32) * B(32x16) + C, bf16 in / fp32 acc.
//  A fragment: lane l holds A[row l&15][k = 8*(l>>4) .. +7]
//  B fragment: lane l holds B[k = 8*(l>>4) .. +7][col l&15]
//  C/D:        lane l, reg r holds D[row 4*(l>>4)+r][col l&15]
__device__ __forceinline__ f32x4 mfma16(bf16x8 a, bf16x8 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
// Sums on DPP (VALU cross-lane operands) instead of __shfl_xor, which compiles to ds_bpermute_b32 and takes the LDS crossbar:
// matters in kernels whose LDS pipe is already busy.
template <int CTRL>
__device__ __forceinline__ float dpp_add_f32(float v) {
    return v + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
// sum over aligned groups of 8 lanes, result in every lane: quad_perm [1,0,3,2], quad_perm [2,3,0,1], row_half_mirror
__device__ __forceinline__ float group8_sum(float v) {
    v = dpp_add_f32<0xB1>(v);
    v = dpp_add_f32<0x4E>(v);
    return dpp_add_f32<0x141>(v);
}
// x[i] (op) x[i ^ 16] and x[i] (op) x[i ^ 32] on v_permlane16_swap / v_permlane32_swap (gfx950): swapping a register with itself
// leaves every lane with its own value in one result and its partner's in the other, so a commutative op of the two is the
// butterfly step -- one VALU instruction instead of the LDS-crossbar round trip of ds_bpermute_b32 that __shfl_xor compiles to.
__device__ __forceinline__ float xor16_sum(float v) {
    const u32x2 r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ float xor32_sum(float v) {
    const u32x2 r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ float xor16_max(float v) {
    const u32x2 r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ float xor32_max(float v) {
    const u32x2 r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}

// v_exp_f32 as is: arguments in the softmax kernels are <= 0 (score minus row maximum / log-sum-exp), results below 2^-126 may flush to zero.
// The library exp2f wraps the same instruction in a denormal-range rescue (compare, select, scale) -- ~5 extra VALU
// instructions per call, a third of the softmax arithmetic of these kernels.
__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }

// erf-form GELU (torch.nn.GELU default).  erf by Abramowitz & Stegun 7.1.26 (|abs err| <= 1.5e-7, far below the
// bf16 rounding of the stored result): one v_rcp, one v_exp and a 5-term Horner chain instead of libm's erff,
// which cost more than the GEMM main loop of the fc1 layers when evaluated 128 times per thread in the epilogue.
__device__ __forceinline__ void erf_parts(float x, float& erf_abs, float& gauss) {   // for z = |x| / sqrt(2)
    const float z = fabsf(x) * 0.70710678118654752f;
    const float t = __builtin_amdgcn_rcpf(1.0f + 0.3275911f * z);
    gauss = __expf(-z * z);                                                        // = exp(-x^2 / 2)
    const float poly = t * (0.254829592f + t * (-0.284496736f + t * (1.421413741f + t * (-1.453152027f + t * 1.061405429f))));
    erf_abs = 1.0f - poly * gauss;
}
__device__ __forceinline__ float gelu_f(float x) {
    float e, g;
    erf_parts(x, e, g);
    return 0.5f * x * (1.0f + copysignf(e, x));
}
__device__ __forceinline__ float dgelu_f(float x) {
    float e, g;
    erf_parts(x, e, g);
    return 0.5f * (1.0f + copysignf(e, x)) + x * g * 0.39894228040143268f;
}

__device__ __forceinline__ bf16x8 cvt8(const float4& a, const float4& b) {
    bf16x8 r;
    r[0] = (bf16)a.x; r[1] = (bf16)a.y; r[2] = (bf16)a.z; r[3] = (bf16)a.w;
    r[4] = (bf16)b.x; r[5] = (bf16)b.y; r[6] = (bf16)b.z; r[7] = (bf16)b.w;
    return r;
}

// XCD-aware remap of a 1-D block id: consecutive hardware block ids are dealt round-robin over
// the 8 XCDs (each with a private L2); give every XCD one contiguous chunk of logical tiles so
// tiles that share an operand panel hit the same L2.  Bijective for any nwg.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7, x = bid & 7;
    const int base = x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q;
    return base + (bid >> 3);
}

// Dropout on attention probabilities (nn.MultiheadAttention(dropout=p) in training mode, reference
// transformer_decoder/transformer.py:249-250): element (b, h, q, key) is KEPT iff hash(seed, index) >= p * 2^32.  A pure function of
// the index, so forward, dQ and dK / dV kernels (and the fp32 exact-mode kernels) regenerate the same mask without storing it.
__device__ __forceinline__ bool attn_keep(unsigned seed, unsigned thresh, unsigned long long idx) {
    unsigned x = ((unsigned)idx ^ ((unsigned)(idx >> 32) * 0x9E3779B9u)) + seed;
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x >= thresh;
}
static inline unsigned attn_drop_thresh(float p) { return p <= 0.f ? 0u : (p >= 1.f ? 0xffffffffu : (unsigned)((double)p * 4294967296.0)); }

static inline int64_t ceil_div64(int64_t a, int64_t b) { return (a + b - 1) / b; }
