// LayerNorm forward / backward over the channel dimension, one 64-lane wave per token row.
//
// Covers norm1 / norm2 / PatchMerging.norm / patch_embed.norm / norm{i} of the Swin backbone
// (reference model/modeling/backbone/swin.py:247, 293, 334, 492, 673) and every post-norm
// "x = LN(x + sublayer(x))" of the pixel decoder and transformer decoder
// (pixel_decoder/msdeformattn.py:136-137, 128-129; transformer_decoder/transformer.py:268-297;
// oneformer_transformer_decoder.py:66-67, 126-127, 184-185), with the residual add fused.
//
// HBM-bound: a row is read once (16-byte loads, kept in registers), statistics by wave shuffles,
// output written once.  Algorithmic bytes per row: C * (in + [res] + out [+ h_out]) element sizes.
#include "common.h"
#include <stdlib.h>

struct LnFwd {
    const void* x; int x_f32;
    const void* res; int res_f32;     // optional residual added before the norm
    float* h_out;                     // optional: x + res (fp32), the tensor that is normalised
    const float* gamma; const float* beta;
    void* y; int y_f32;
    bf16* y16;                        // optional bf16 twin of y (the next GEMM's operand) when y is fp32
    float2* stats;                    // optional (mean, rstd) per row
    long M; int C; float eps;
    int mgH, mgW, mgC;                // > 0: PatchMerging gather (see PatchGather): x is the (B, mgH, mgW, mgC) fp32 map, C == 4 * mgC
};

// PatchMerging (reference backbone/swin.py:311-331): output token (b, i, j) of the (ceil(H/2), ceil(W/2)) grid is the concatenation
// [x(2i, 2j) | x(2i+1, 2j) | x(2i, 2j+1) | x(2i+1, 2j+1)] of four C-channel pixels, zeros where H or W is odd.  Folded into the
// LayerNorm's row addressing instead of materialising the gathered (B, L/4, 4C) tensor: column c of a row lives at
// base(b, i, j) + off(c) when its pixel is inside the map.
struct PatchGather {
    long base; bool ok[4];
    __device__ __forceinline__ PatchGather(long row, int H, int W, int C) {
        const int W2 = (W + 1) >> 1, H2 = (H + 1) >> 1;
        const int j = (int)(row % W2);
        const long r2 = row / W2;
        const int i = (int)(r2 % H2);
        const long b = r2 / H2;
        base = ((b * H + 2 * i) * W + 2 * j) * C;
        const bool y1 = 2 * i + 1 < H, x1 = 2 * j + 1 < W;
        ok[0] = true; ok[1] = y1; ok[2] = x1; ok[3] = y1 && x1;
    }
};
// column c (multiple of 4) of a gathered row: segment (c / C) -> pixel (dy = seg & 1, dx = seg >> 1), channel c % C
__device__ __forceinline__ void patch_col(int c, int W, int C, int& seg, long& off) {
    seg = c / C;
    off = ((long)(seg & 1) * W + (seg >> 1)) * C + (c - seg * C);
}

__device__ __forceinline__ float4 load4(const void* base, int is_f32, long idx) {
    if (is_f32) return *(const float4*)((const float*)base + idx);
    const bf16x4 v = *(const bf16x4*)((const bf16*)base + idx);
    return make_float4((float)v[0], (float)v[1], (float)v[2], (float)v[3]);
}
__device__ __forceinline__ void store4(void* base, int is_f32, long idx, float4 v) {
    if (is_f32) { *(float4*)((float*)base + idx) = v; return; }
    bf16x4 o; o[0] = (bf16)v.x; o[1] = (bf16)v.y; o[2] = (bf16)v.z; o[3] = (bf16)v.w;
    *(bf16x4*)((bf16*)base + idx) = o;
}

template <int NV, bool MG>
__global__ __launch_bounds__(256) void ln_fwd_kernel(LnFwd p) {
    const int lane = threadIdx.x & 63;
    const long wave0 = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const long nwaves = (long)gridDim.x * 4;
    const float invC = 1.0f / (float)p.C;
    int gseg[NV];
    long goff[NV];
    if (MG) {
#pragma unroll
        for (int i = 0; i < NV; ++i) patch_col(min(lane * 4 + i * 256, p.C - 4), p.mgW, p.mgC, gseg[i], goff[i]);
    }
    for (long row = wave0; row < p.M; row += nwaves) {
        float4 v[NV];
        float s = 0.f;
        const PatchGather pg(MG ? row : 0, MG ? p.mgH : 1, MG ? p.mgW : 1, p.mgC);
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = lane * 4 + i * 256;
            if (c < p.C) {
                if (MG) v[i] = pg.ok[gseg[i]] ? *(const float4*)((const float*)p.x + pg.base + goff[i]) : make_float4(0.f, 0.f, 0.f, 0.f);
                else v[i] = load4(p.x, p.x_f32, row * p.C + c);
                if (p.res != nullptr) {
                    const float4 r = load4(p.res, p.res_f32, row * p.C + c);
                    v[i].x += r.x; v[i].y += r.y; v[i].z += r.z; v[i].w += r.w;
                }
                if (p.h_out != nullptr) *(float4*)(p.h_out + row * p.C + c) = v[i];
                s += v[i].x + v[i].y + v[i].z + v[i].w;
            } else {
                v[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
        const float mean = wave_sum(s) * invC;
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = lane * 4 + i * 256;
            if (c < p.C) {
                const float a = v[i].x - mean, b = v[i].y - mean, cc = v[i].z - mean, d = v[i].w - mean;
                q += a * a + b * b + cc * cc + d * d;
            }
        }
        const float rstd = rsqrtf(wave_sum(q) * invC + p.eps);
        if (p.stats != nullptr && lane == 0) p.stats[row] = make_float2(mean, rstd);
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = lane * 4 + i * 256;
            if (c < p.C) {
                const float4 g = *(const float4*)(p.gamma + c), b = *(const float4*)(p.beta + c);
                float4 o;
                o.x = (v[i].x - mean) * rstd * g.x + b.x;
                o.y = (v[i].y - mean) * rstd * g.y + b.y;
                o.z = (v[i].z - mean) * rstd * g.z + b.z;
                o.w = (v[i].w - mean) * rstd * g.w + b.w;
                store4(p.y, p.y_f32, row * p.C + c, o);
                if (p.y16 != nullptr) store4(p.y16, 0, row * p.C + c, o);
            }
        }
    }
}

// the PatchMerging gather is a template flag: the plain LayerNorms keep their register budgets (and occupancy)
#define LN_LAUNCH(DIR, NVV, GRID, BLOCK, SHM, STREAM, P)                                                          \
    do {                                                                                                          \
        if ((P).mgC > 0) hipLaunchKernelGGL((ln_##DIR##_kernel<NVV, true>), GRID, BLOCK, SHM, STREAM, P);        \
        else hipLaunchKernelGGL((ln_##DIR##_kernel<NVV, false>), GRID, BLOCK, SHM, STREAM, P);                   \
    } while (0)

static int ln_fwd_launch(LnFwd& p, hipStream_t stream);

extern "C" int uenc_layernorm_fwd(const void* x, int x_dtype, const void* res, int res_dtype, float* h_out,
                                  const float* gamma, const float* beta, void* y, int y_dtype, float* stats,
                                  long M, int C, float eps, void* y16, hipStream_t stream) {
    UENC_CHECK_ARG(x && gamma && beta && y && M > 0 && C > 0 && C % 4 == 0 && C <= 6144);
    LnFwd p;
    p.x = x; p.x_f32 = (x_dtype == UENC_F32); p.res = res; p.res_f32 = (res_dtype == UENC_F32);
    p.h_out = h_out; p.gamma = gamma; p.beta = beta; p.y = y; p.y_f32 = (y_dtype == UENC_F32); p.y16 = (bf16*)y16;
    p.stats = (float2*)stats; p.M = M; p.C = C; p.eps = eps; p.mgH = p.mgW = p.mgC = 0;
    return ln_fwd_launch(p, stream);
}

static int ln_fwd_launch(LnFwd& p, hipStream_t stream) {
    const long M = p.M; const int C = p.C;
    long blocks = (M + 3) / 4;
    if (blocks > 8192) blocks = 8192;
    const int nv = (C + 255) / 256;
    dim3 grid((unsigned)blocks), block(256);
    if (nv <= 1) LN_LAUNCH(fwd, 1, grid, block, 0, stream, p);
    else if (nv <= 2) LN_LAUNCH(fwd, 2, grid, block, 0, stream, p);
    else if (nv <= 4) LN_LAUNCH(fwd, 4, grid, block, 0, stream, p);
    else if (nv <= 8) LN_LAUNCH(fwd, 8, grid, block, 0, stream, p);
    else if (nv <= 16) LN_LAUNCH(fwd, 16, grid, block, 0, stream, p);
    else LN_LAUNCH(fwd, 24, grid, block, 0, stream, p);
    UENC_LAUNCH_RET();
}

// backward:  xhat = (h - mean) * rstd ;  g = dy * gamma
//   dx = rstd * (g - mean_c(g) - xhat * mean_c(g * xhat)) [+ dres]
//   dgamma += sum_rows dy * xhat ; dbeta += sum_rows dy
struct LnBwd {
    const void* dy; int dy_f32;
    const void* h; int h_f32;
    const float2* stats;
    const float* gamma;
    const float* dres;               // optional fp32 gradient arriving on the skip path
    void* dx; int dx_f32;
    bf16* dx16;                      // optional bf16 twin of dx (operand of the GEMMs that consume the gradient)
    float* dgamma; float* dbeta;     // accumulated
    float* part;                     // optional (gridDim.x, 2, C) block partials of dgamma / dbeta (then summed by ln_bwd_param_kernel)
    long M; int C;
    int mgH, mgW, mgC;               // > 0: PatchMerging: h is the (B, mgH, mgW, mgC) map gathered per row, dx is scattered back to it
};

template <int NV, bool MG>
__global__ __launch_bounds__(256) void ln_bwd_kernel(LnBwd p) {
    extern __shared__ __attribute__((aligned(16))) float sh[];   // [4 waves][2][C] partials of dgamma / dbeta
    const int lane = threadIdx.x & 63;
    const long wave0 = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const long nwaves = (long)gridDim.x * 4;
    const float invC = 1.0f / (float)p.C;
    float4 ag[NV], ab[NV], gm[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        ag[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        ab[i] = ag[i];
        const int c = lane * 4 + i * 256;
        gm[i] = c < p.C ? *(const float4*)(p.gamma + c) : ag[i];
    }
    int gseg[NV];
    long goff[NV];
    if (MG) {
#pragma unroll
        for (int i = 0; i < NV; ++i) patch_col(min(lane * 4 + i * 256, p.C - 4), p.mgW, p.mgC, gseg[i], goff[i]);
    }
    for (long row = wave0; row < p.M; row += nwaves) {
        const float2 st = p.stats[row];
        float4 d[NV], xh[NV], rs[NV];
        float s1 = 0.f, s2 = 0.f;
        const PatchGather pg(MG ? row : 0, MG ? p.mgH : 1, MG ? p.mgW : 1, p.mgC);
        // the skip-path gradient is fetched together with dy and h: all of a row's loads are in flight at once (issued behind the
        // row reductions they added a second full memory latency to every row)
        if (p.dres != nullptr) {
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                const int c = lane * 4 + i * 256;
                if (c < p.C) rs[i] = *(const float4*)(p.dres + row * p.C + c);
            }
        }
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = lane * 4 + i * 256;
            if (c < p.C) {
                d[i] = load4(p.dy, p.dy_f32, row * p.C + c);
                float4 hv;
                if (MG) hv = pg.ok[gseg[i]] ? *(const float4*)((const float*)p.h + pg.base + goff[i]) : make_float4(0.f, 0.f, 0.f, 0.f);
                else hv = load4(p.h, p.h_f32, row * p.C + c);
                xh[i] = make_float4((hv.x - st.x) * st.y, (hv.y - st.x) * st.y, (hv.z - st.x) * st.y, (hv.w - st.x) * st.y);
                ag[i].x += d[i].x * xh[i].x; ag[i].y += d[i].y * xh[i].y; ag[i].z += d[i].z * xh[i].z; ag[i].w += d[i].w * xh[i].w;
                ab[i].x += d[i].x; ab[i].y += d[i].y; ab[i].z += d[i].z; ab[i].w += d[i].w;
                d[i].x *= gm[i].x; d[i].y *= gm[i].y; d[i].z *= gm[i].z; d[i].w *= gm[i].w;
                s1 += d[i].x + d[i].y + d[i].z + d[i].w;
                s2 += d[i].x * xh[i].x + d[i].y * xh[i].y + d[i].z * xh[i].z + d[i].w * xh[i].w;
            }
        }
        s1 = wave_sum(s1) * invC;
        s2 = wave_sum(s2) * invC;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = lane * 4 + i * 256;
            if (c < p.C) {
                float4 o;
                o.x = st.y * (d[i].x - s1 - xh[i].x * s2);
                o.y = st.y * (d[i].y - s1 - xh[i].y * s2);
                o.z = st.y * (d[i].z - s1 - xh[i].z * s2);
                o.w = st.y * (d[i].w - s1 - xh[i].w * s2);
                if (p.dres != nullptr) { o.x += rs[i].x; o.y += rs[i].y; o.z += rs[i].z; o.w += rs[i].w; }
                if (MG) {
                    if (pg.ok[gseg[i]]) *(float4*)((float*)p.dx + pg.base + goff[i]) = o;       // every map pixel is written exactly once
                } else {
                    store4(p.dx, p.dx_f32, row * p.C + c, o);
                    if (p.dx16 != nullptr) store4(p.dx16, 0, row * p.C + c, o);
                }
            }
        }
    }
    if (p.dgamma != nullptr && p.C > 2048) {            // wide rows: one [2][C] slab, LDS atomics (4 slabs would not fit)
        for (int c = threadIdx.x; c < 2 * p.C; c += 256) sh[c] = 0.f;
        __syncthreads();
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = lane * 4 + i * 256;
            if (c < p.C) {
                atomicAdd(&sh[c + 0], ag[i].x); atomicAdd(&sh[c + 1], ag[i].y);
                atomicAdd(&sh[c + 2], ag[i].z); atomicAdd(&sh[c + 3], ag[i].w);
                atomicAdd(&sh[p.C + c + 0], ab[i].x); atomicAdd(&sh[p.C + c + 1], ab[i].y);
                atomicAdd(&sh[p.C + c + 2], ab[i].z); atomicAdd(&sh[p.C + c + 3], ab[i].w);
            }
        }
        __syncthreads();
        for (int c = threadIdx.x; c < 2 * p.C; c += 256) {
            if (p.part != nullptr) p.part[(long)blockIdx.x * 2 * p.C + c] = sh[c];
            else atomicAdd((c < p.C ? p.dgamma + c : p.dbeta + (c - p.C)), sh[c]);
        }
    } else if (p.dgamma != nullptr) {
        // block partial: each wave parks its sums in its own LDS slab (plain 16-byte stores; LDS float atomics cost ~3 cycles
        // per lane), the slabs are added while being written out
        const int wv = threadIdx.x >> 6;
        float* mine = sh + (long)wv * 2 * p.C;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = lane * 4 + i * 256;
            if (c < p.C) { *(float4*)(mine + c) = ag[i]; *(float4*)(mine + p.C + c) = ab[i]; }
        }
        __syncthreads();
        for (int c = threadIdx.x; c < 2 * p.C; c += 256) {
            const float v = sh[c] + sh[2 * p.C + c] + sh[4 * p.C + c] + sh[6 * p.C + c];
            if (p.part != nullptr) p.part[(long)blockIdx.x * 2 * p.C + c] = v;          // one small kernel sums the blocks
            else atomicAdd((c < p.C ? p.dgamma + c : p.dbeta + (c - p.C)), v);
        }
    }
}

// dgamma[c] += sum_b part[b][0][c], dbeta[c] += sum_b part[b][1][c]: every float atomic is a 64-byte transaction at the
// memory side, and nblk x 2C of them per LayerNorm backward cost more than the LayerNorm itself (1024 x 1536 -> 77 us).
__global__ __launch_bounds__(256) void ln_bwd_param_kernel(const float* __restrict__ part, int nblk, int C, float* __restrict__ dgamma,
                                                           float* __restrict__ dbeta) {
    // grid (column groups of 64, 16 slices of the partial rows): 16 atomics per parameter instead of nblk
    __shared__ float red[4][64];
    const int col = blockIdx.x * 64 + (threadIdx.x & 63), r = threadIdx.x >> 6;      // col over [0, 2C)
    const int per = (nblk + gridDim.y - 1) / gridDim.y, b0 = blockIdx.y * per, b1 = min(nblk, b0 + per);
    float acc = 0.f;
    if (col < 2 * C)
        for (int b = b0 + r; b < b1; b += 4) acc += part[(long)b * 2 * C + col];
    red[r][threadIdx.x & 63] = acc;
    __syncthreads();
    if (r == 0 && col < 2 * C) {
        const float v = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
        atomicAdd(col < C ? dgamma + col : dbeta + (col - C), v);
    }
}

static int ln_bwd_launch(LnBwd& p, float* part_ws, int defer, hipStream_t stream);

extern "C" int uenc_layernorm_bwd(const void* dy, int dy_dtype, const void* h, int h_dtype, const float* stats,
                                  const float* gamma, const float* dres, void* dx, int dx_dtype, float* dgamma,
                                  float* dbeta, long M, int C, void* dx16, float* part_ws, int defer_param_sums, hipStream_t stream) {
    UENC_CHECK_ARG(dy && h && stats && gamma && dx && M > 0 && C > 0 && C % 4 == 0 && C <= 6144);
    UENC_CHECK_ARG((dgamma == nullptr) == (dbeta == nullptr));
    LnBwd p;
    p.dy = dy; p.dy_f32 = (dy_dtype == UENC_F32); p.h = h; p.h_f32 = (h_dtype == UENC_F32);
    p.stats = (const float2*)stats; p.gamma = gamma; p.dres = dres; p.dx = dx; p.dx_f32 = (dx_dtype == UENC_F32); p.dx16 = (bf16*)dx16;
    p.dgamma = dgamma; p.dbeta = dbeta; p.M = M; p.C = C; p.mgH = p.mgW = p.mgC = 0;
    return ln_bwd_launch(p, part_ws, defer_param_sums, stream);
}

// workgroups of a LayerNorm backward over M rows, and whether their dgamma / dbeta partials are STORED (then summed by a second kernel)
static long ln_bwd_blocks(long M, int C, bool with_ws, bool* stored) {
    long blocks = (M + 3) / 4;
    long cap = with_ws ? 2048 : 1024;      // part_ws: (2048, 2, C) floats
    { const char* e = getenv("UENC_LN_BWD_BLOCKS"); if (e && atoi(e) >= 64 && atoi(e) <= 2048) cap = atoi(e); }     // tuning knob
    if (blocks > cap) blocks = cap;
    // few rows (the decoder's 300-token LayerNorms): the handful of block partials goes straight to dgamma / dbeta by atomics,
    // a second launch would cost more than it saves
    *stored = with_ws && blocks * 2 * C > 131072;
    return blocks;
}

// Rows of stored partials ([nblk][2][C] floats) a LayerNorm backward over (M, C) with a scratch buffer writes; 0: it adds its few block sums
// atomically and nothing is left to reduce.  For callers that defer the reduction (defer_param_sums) and run uenc_ln_param_grouped later.
extern "C" int uenc_layernorm_bwd_blocks(long M, int C) {
    if (M <= 0 || C <= 0) return 0;
    bool stored = false;
    const long b = ln_bwd_blocks(M, C, true, &stored);
    return stored ? (int)b : 0;
}

static int ln_bwd_launch(LnBwd& p, float* part_ws, int defer, hipStream_t stream) {
    const long M = p.M; const int C = p.C;
    float* dgamma = p.dgamma; float* dbeta = p.dbeta;
    bool stored = false;
    const long blocks = ln_bwd_blocks(M, C, part_ws != nullptr && dgamma != nullptr, &stored);
    p.part = stored ? part_ws : nullptr;
    const int nv = (C + 255) / 256;
    const size_t shm = (size_t)(C > 2048 ? 1 : 4) * 2 * C * sizeof(float);      // <= 64 KB
    dim3 grid((unsigned)blocks), block(256);
    if (nv <= 1) LN_LAUNCH(bwd, 1, grid, block, shm, stream, p);
    else if (nv <= 2) LN_LAUNCH(bwd, 2, grid, block, shm, stream, p);
    else if (nv <= 4) LN_LAUNCH(bwd, 4, grid, block, shm, stream, p);
    else if (nv <= 8) LN_LAUNCH(bwd, 8, grid, block, shm, stream, p);
    else if (nv <= 16) LN_LAUNCH(bwd, 16, grid, block, shm, stream, p);
    else LN_LAUNCH(bwd, 24, grid, block, shm, stream, p);
    if (p.part != nullptr && !defer)
        hipLaunchKernelGGL(ln_bwd_param_kernel, dim3((2 * C + 63) / 64, 16), dim3(256), 0, stream, (const float*)p.part, (int)blocks, C, dgamma, dbeta);
    UENC_LAUNCH_RET();
}

// Grouped form of ln_bwd_param_kernel: the dgamma / dbeta sums of MANY LayerNorm backward passes in one launch.  A backward pass leaves
// ~70 of these reductions (a few MB of block partials each); alone each is a 10 us launch for ~2 us of work.  Their results are needed
// only by the optimiser, so the callers park the partials (uenc_layernorm_bwd(..., defer_param_sums = 1)) and run this once.
// table: n descriptors in device memory, 40 bytes each: { const float* part; float* dgamma; float* dbeta; int nblk, C, group_begin, pad; }
// group_begin = exclusive prefix sum of ceil(2C / 64); total_groups = the full sum.
struct LnParamDesc { const float* part; float* dgamma; float* dbeta; int nblk, C, group_begin, pad; };
__global__ __launch_bounds__(256) void ln_param_grouped_kernel(const LnParamDesc* __restrict__ table, int n) {
    __shared__ float red[4][64];
    int lo = 0, hi = n - 1;
    const int grp = blockIdx.x;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (table[mid].group_begin <= grp) lo = mid; else hi = mid - 1;
    }
    const LnParamDesc d = table[lo];
    const int C = d.C, nblk = d.nblk;
    const int col = (grp - d.group_begin) * 64 + (threadIdx.x & 63), r = threadIdx.x >> 6;      // col over [0, 2C)
    const int per = (nblk + gridDim.y - 1) / gridDim.y, b0 = blockIdx.y * per, b1 = min(nblk, b0 + per);
    float acc = 0.f;
    if (col < 2 * C)
        for (int b = b0 + r; b < b1; b += 4) acc += d.part[(long)b * 2 * C + col];
    red[r][threadIdx.x & 63] = acc;
    __syncthreads();
    if (r == 0 && col < 2 * C) {
        const float v = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
        atomicAdd(col < C ? d.dgamma + col : d.dbeta + (col - C), v);
    }
}

extern "C" int uenc_ln_param_grouped(const void* table, int n, int total_groups, hipStream_t stream) {
    UENC_CHECK_ARG(table && n > 0 && total_groups > 0 && ((uintptr_t)table & 7) == 0);
    static_assert(sizeof(LnParamDesc) == 40, "descriptor layout is part of the ABI");
    hipLaunchKernelGGL(ln_param_grouped_kernel, dim3((unsigned)total_groups, 16), dim3(256), 0, stream, (const LnParamDesc*)table, n);
    UENC_LAUNCH_RET();
}


// ---- PatchMerging gather + LayerNorm(4C) as one pass each way (reference backbone/swin.py:311-334: pad to even, four strided
// slices, cat, norm; the 4C -> 2C reduction GEMM follows).  x (B, H, W, C) fp32 -> y (B * ceil(H/2) * ceil(W/2), 4C) bf16;
// backward: dy (same rows, bf16 | fp32) -> dx (B, H, W, C) fp32, every element written once (no zeroing needed). ----
extern "C" int uenc_patch_merge_ln_fwd(const float* x, const float* gamma, const float* beta, void* y, float* stats, int B, int H, int W,
                                       int C, float eps, hipStream_t stream) {
    UENC_CHECK_ARG(x && gamma && beta && y && stats && B > 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0 && 4 * C <= 6144);
    UENC_CHECK_ARG((((uintptr_t)x | (uintptr_t)y) & 15) == 0);
    LnFwd p;
    p.x = x; p.x_f32 = 1; p.res = nullptr; p.res_f32 = 0; p.h_out = nullptr; p.gamma = gamma; p.beta = beta; p.y = y; p.y_f32 = 0;
    p.y16 = nullptr; p.stats = (float2*)stats; p.M = (long)B * ((H + 1) / 2) * ((W + 1) / 2); p.C = 4 * C; p.eps = eps;
    p.mgH = H; p.mgW = W; p.mgC = C;
    return ln_fwd_launch(p, stream);
}

extern "C" int uenc_patch_merge_ln_bwd(const void* dy, int dy_dtype, const float* x, const float* stats, const float* gamma, float* dx,
                                       float* dgamma, float* dbeta, float* part_ws, int B, int H, int W, int C, int defer_param_sums,
                                       hipStream_t stream) {
    UENC_CHECK_ARG(dy && x && stats && gamma && dx && B > 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0 && 4 * C <= 6144);
    UENC_CHECK_ARG((dgamma == nullptr) == (dbeta == nullptr));
    LnBwd p;
    p.dy = dy; p.dy_f32 = (dy_dtype == UENC_F32); p.h = x; p.h_f32 = 1; p.stats = (const float2*)stats; p.gamma = gamma; p.dres = nullptr;
    p.dx = dx; p.dx_f32 = 1; p.dx16 = nullptr; p.dgamma = dgamma; p.dbeta = dbeta;
    p.M = (long)B * ((H + 1) / 2) * ((W + 1) / 2); p.C = 4 * C; p.mgH = H; p.mgW = W; p.mgC = C;
    return ln_bwd_launch(p, part_ws, defer_param_sums, stream);
}
