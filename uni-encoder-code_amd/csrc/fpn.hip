// Channels-last pieces of the pixel decoder's FPN branch (gfx950): GroupNorm forward / backward on token matrices, the
// bilinear top-down merge, and im2col / col2im of the 3x3 output convolution.
//
// Reference: model/modeling/pixel_decoder/msdeformattn.py:283-302 (lateral 1x1 conv + GN, output 3x3 conv + GN + ReLU),
// :343-352 (cur_fpn + F.interpolate(out[-1], size, "bilinear", align_corners=False)) and the mask_features 1x1 conv :304.
// The reference runs these in NCHW; here every map of the branch stays a token matrix (B, H*W, C) -- the layout the
// GEMMs before and after produce and consume -- so none of the 268 MB maps at 1/4 resolution is ever transposed.
// All kernels are HBM-bound streaming passes: lane = 4 consecutive channels (16 bytes of fp32), a 256-channel token is
// one coalesced 1 KB row per wave instruction.
//
// GroupNorm over (H*W x C/G) per (image, group), as torch.nn.GroupNorm: biased variance, eps inside the square root.
//   forward : pass 1 per-(image, chunk of tokens, channel) sums of x and x^2 in fp64 (no cancellation issue, fp64 adds are
//             far from being the bottleneck of a streaming pass), pass 2 (tiny) mean / rstd per (image, group), pass 3
//             y = (x - mean) * rstd * gamma + beta [+ bilinear(prev)] [ReLU];
//   backward: per-(image, channel) sums of dy and dy * xhat (they give dgamma / dbeta AND, weighted by gamma, the two group
//             means the input gradient needs), then dx = rstd * (dy * gamma - c1 - xhat * c2).
#include "common.h"

#define GN_THREADS 256

struct GnP {
    const void* x; int x_f32;            // (B, HW, C) GroupNorm input
    const float* gamma; const float* beta;
    void* y; int y_f32;                  // forward output / backward dx
    const void* dy; int dy_f32;          // backward: gradient of the output
    float* stats;                        // (B, G, 2) mean, rstd
    double* part;                        // (B, nchunk, C, 2) partial sums
    float* coef;                         // backward: (B, G, 2) c1, c2
    float* dgamma; float* dbeta;         // accumulated (+=)
    const float* add_src;                // forward: optional (B, Hs, Ws, C) fp32 map merged in by bilinear resize
    int B, HW, C, G, nchunk, relu;
    int H, W, Hs, Ws;
    float eps;
};

__device__ __forceinline__ float4 ld4(const void* base, int is_f32, long idx) {
    if (is_f32) return *(const float4*)((const float*)base + idx);
    const bf16x4 v = *(const bf16x4*)((const bf16*)base + idx);
    return make_float4((float)v[0], (float)v[1], (float)v[2], (float)v[3]);
}
__device__ __forceinline__ void st4(void* base, int is_f32, long idx, const float4& v) {
    if (is_f32) { *(float4*)((float*)base + idx) = v; return; }
    bf16x4 o; o[0] = (bf16)v.x; o[1] = (bf16)v.y; o[2] = (bf16)v.z; o[3] = (bf16)v.w;
    *(bf16x4*)((bf16*)base + idx) = o;
}

// pass 1 of forward (MODE 0: sums of x, x^2) and of backward (MODE 1: sums of dy, dy * xhat, ReLU mask applied to dy).
// grid (nchunk, B); thread = (channel quad cq = t % (C/4), token row t / (C/4)).
template <int MODE>
__global__ __launch_bounds__(GN_THREADS) void gn_reduce_kernel(GnP p) {
    __shared__ double red[GN_THREADS][8];
    const int cqn = p.C >> 2, t = threadIdx.x, cq = t % cqn, trow = t / cqn, nrow = GN_THREADS / cqn;
    const int b = blockIdx.y, chunk = blockIdx.x;
    const int per = (p.HW + p.nchunk - 1) / p.nchunk;
    const int t0 = chunk * per, t1 = min(p.HW, t0 + per);
    const int c = cq * 4;
    double a[4] = {0, 0, 0, 0}, q[4] = {0, 0, 0, 0};
    float mean = 0.f, rstd = 0.f;
    float4 ga = make_float4(0.f, 0.f, 0.f, 0.f), be = ga;
    if (MODE == 1) {
        const int g = c / (p.C / p.G);
        mean = p.stats[((long)b * p.G + g) * 2]; rstd = p.stats[((long)b * p.G + g) * 2 + 1];
        ga = *(const float4*)(p.gamma + c); be = *(const float4*)(p.beta + c);
    }
    for (int tok = t0 + trow; tok < t1; tok += nrow) {
        const long idx = ((long)b * p.HW + tok) * p.C + c;
        const float4 x = ld4(p.x, p.x_f32, idx);
        if (MODE == 0) {
            a[0] += x.x; a[1] += x.y; a[2] += x.z; a[3] += x.w;
            q[0] += (double)x.x * x.x; q[1] += (double)x.y * x.y; q[2] += (double)x.z * x.z; q[3] += (double)x.w * x.w;
        } else {
            float4 d = ld4(p.dy, p.dy_f32, idx);
            const float4 xh = make_float4((x.x - mean) * rstd, (x.y - mean) * rstd, (x.z - mean) * rstd, (x.w - mean) * rstd);
            if (p.relu) {
                if (xh.x * ga.x + be.x <= 0.f) d.x = 0.f;
                if (xh.y * ga.y + be.y <= 0.f) d.y = 0.f;
                if (xh.z * ga.z + be.z <= 0.f) d.z = 0.f;
                if (xh.w * ga.w + be.w <= 0.f) d.w = 0.f;
            }
            a[0] += d.x; a[1] += d.y; a[2] += d.z; a[3] += d.w;
            q[0] += (double)d.x * xh.x; q[1] += (double)d.y * xh.y; q[2] += (double)d.z * xh.z; q[3] += (double)d.w * xh.w;
        }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) { red[t][j] = a[j]; red[t][4 + j] = q[j]; }
    __syncthreads();
    if (trow == 0) {
        for (int r = 1; r < nrow; ++r)
#pragma unroll
            for (int j = 0; j < 8; ++j) red[t][j] += red[t + r * cqn][j];
        double* out = p.part + (((long)b * p.nchunk + chunk) * p.C + c) * 2;
#pragma unroll
        for (int j = 0; j < 4; ++j) { out[2 * j] = red[t][j]; out[2 * j + 1] = red[t][4 + j]; }
    }
}

__device__ __forceinline__ double wave_sum_f64(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// pass 2 of forward: one wave per (image, group): mean and rstd from the fp64 partials
__global__ __launch_bounds__(64) void gn_stats_kernel(GnP p) {
    const int i = blockIdx.x, b = i / p.G, g = i - b * p.G, cpg = p.C / p.G;
    double s = 0, ss = 0;
    for (int e = threadIdx.x; e < p.nchunk * cpg; e += 64) {
        const int ch = e / cpg, j = e - ch * cpg;
        const double* q = p.part + (((long)b * p.nchunk + ch) * p.C + g * cpg + j) * 2;
        s += q[0]; ss += q[1];
    }
    s = wave_sum_f64(s); ss = wave_sum_f64(ss);
    if (threadIdx.x == 0) {
        const double n = (double)p.HW * cpg, mean = s / n;
        double var = ss / n - mean * mean;
        if (var < 0) var = 0;
        p.stats[(long)i * 2] = (float)mean;
        p.stats[(long)i * 2 + 1] = (float)(1.0 / sqrt(var + (double)p.eps));
    }
}

// pass 2 of backward: one workgroup per (image, 64 channels): per-channel totals -> dgamma / dbeta (+=, atomics over images)
// and, for the groups inside these 64 channels, c1 = mean_g(dy gamma), c2 = mean_g(dy gamma xhat)
__global__ __launch_bounds__(GN_THREADS) void gn_bwd_coef_kernel(GnP p) {
    __shared__ double sa[4][64], sq[4][64];
    const int nblk = (p.C + 63) / 64, b = blockIdx.x / nblk, c0 = (blockIdx.x - b * nblk) * 64;
    const int cl = threadIdx.x & 63, r = threadIdx.x >> 6, c = c0 + cl;
    double a = 0, q = 0;
    if (c < p.C)
        for (int ch = r; ch < p.nchunk; ch += 4) {
            const double* e = p.part + (((long)b * p.nchunk + ch) * p.C + c) * 2;
            a += e[0]; q += e[1];
        }
    sa[r][cl] = a; sq[r][cl] = q;
    __syncthreads();
    if (r == 0 && c < p.C) {
        a = sa[0][cl] + sa[1][cl] + sa[2][cl] + sa[3][cl];
        q = sq[0][cl] + sq[1][cl] + sq[2][cl] + sq[3][cl];
        if (p.dbeta != nullptr) atomicAdd(p.dbeta + c, (float)a);
        if (p.dgamma != nullptr) atomicAdd(p.dgamma + c, (float)q);
        const double gm = (double)p.gamma[c];
        sa[0][cl] = a * gm; sq[0][cl] = q * gm;
    }
    __syncthreads();
    const int cpg = p.C / p.G;                       // divides 64 (checked by the launcher)
    if (threadIdx.x < 64 / cpg && c0 + threadIdx.x * cpg < p.C) {
        double s1 = 0, s2 = 0;
        for (int j = 0; j < cpg; ++j) { s1 += sa[0][threadIdx.x * cpg + j]; s2 += sq[0][threadIdx.x * cpg + j]; }
        const double n = (double)p.HW * cpg;
        const int g = c0 / cpg + threadIdx.x;
        p.coef[((long)b * p.G + g) * 2] = (float)(s1 / n);
        p.coef[((long)b * p.G + g) * 2 + 1] = (float)(s2 / n);
    }
}

// bilinear tap of a channels-last fp32 map (align_corners = False, as F.interpolate)
__device__ __forceinline__ float4 bilinear4(const float* src, long img_base, int Hs, int Ws, int C, int c, int oy, int ox, float sy, float sx) {
    float fy = ((float)oy + 0.5f) * sy - 0.5f, fx = ((float)ox + 0.5f) * sx - 0.5f;
    fy = fy < 0.f ? 0.f : fy; fx = fx < 0.f ? 0.f : fx;
    const int y0 = min((int)fy, Hs - 1), y1 = min(y0 + 1, Hs - 1), x0 = min((int)fx, Ws - 1), x1 = min(x0 + 1, Ws - 1);
    const float ly = fy - (float)y0, lx = fx - (float)x0, hy = 1.f - ly, hx = 1.f - lx;
    const float4 a = *(const float4*)(src + img_base + ((long)y0 * Ws + x0) * C + c);
    const float4 b = *(const float4*)(src + img_base + ((long)y0 * Ws + x1) * C + c);
    const float4 d = *(const float4*)(src + img_base + ((long)y1 * Ws + x0) * C + c);
    const float4 e = *(const float4*)(src + img_base + ((long)y1 * Ws + x1) * C + c);
    return make_float4(hy * (hx * a.x + lx * b.x) + ly * (hx * d.x + lx * e.x), hy * (hx * a.y + lx * b.y) + ly * (hx * d.y + lx * e.y),
                       hy * (hx * a.z + lx * b.z) + ly * (hx * d.z + lx * e.z), hy * (hx * a.w + lx * b.w) + ly * (hx * d.w + lx * e.w));
}

// pass 3 of forward (BWD = false) / of backward (BWD = true): one thread per 4 channels of a token
template <bool BWD>
__global__ __launch_bounds__(GN_THREADS) void gn_apply_kernel(GnP p) {
    const int cqn = p.C >> 2, cpg = p.C / p.G;
    const long total = (long)p.B * p.HW * cqn;
    const float sy = p.add_src ? (float)p.Hs / (float)p.H : 0.f, sx = p.add_src ? (float)p.Ws / (float)p.W : 0.f;
    for (long i = (long)blockIdx.x * GN_THREADS + threadIdx.x; i < total; i += (long)gridDim.x * GN_THREADS) {
        const int cq = (int)(i % cqn);
        const long tokg = i / cqn;
        const int b = (int)(tokg / p.HW), c = cq * 4, g = c / cpg;
        const long idx = tokg * p.C + c;
        const float mean = p.stats[((long)b * p.G + g) * 2], rstd = p.stats[((long)b * p.G + g) * 2 + 1];
        const float4 ga = *(const float4*)(p.gamma + c);
        const float4 x = ld4(p.x, p.x_f32, idx);
        const float4 xh = make_float4((x.x - mean) * rstd, (x.y - mean) * rstd, (x.z - mean) * rstd, (x.w - mean) * rstd);
        float4 o;
        if (!BWD) {
            const float4 be = *(const float4*)(p.beta + c);
            o = make_float4(xh.x * ga.x + be.x, xh.y * ga.y + be.y, xh.z * ga.z + be.z, xh.w * ga.w + be.w);
            if (p.add_src) {
                const int tok = (int)(tokg - (long)b * p.HW);
                const float4 u = bilinear4(p.add_src, (long)b * p.Hs * p.Ws * p.C, p.Hs, p.Ws, p.C, c, tok / p.W, tok % p.W, sy, sx);
                o.x += u.x; o.y += u.y; o.z += u.z; o.w += u.w;
            }
            if (p.relu) { o.x = fmaxf(o.x, 0.f); o.y = fmaxf(o.y, 0.f); o.z = fmaxf(o.z, 0.f); o.w = fmaxf(o.w, 0.f); }
        } else {
            float4 d = ld4(p.dy, p.dy_f32, idx);
            if (p.relu) {
                const float4 be = *(const float4*)(p.beta + c);
                if (xh.x * ga.x + be.x <= 0.f) d.x = 0.f;
                if (xh.y * ga.y + be.y <= 0.f) d.y = 0.f;
                if (xh.z * ga.z + be.z <= 0.f) d.z = 0.f;
                if (xh.w * ga.w + be.w <= 0.f) d.w = 0.f;
            }
            const float c1 = p.coef[((long)b * p.G + g) * 2], c2 = p.coef[((long)b * p.G + g) * 2 + 1];
            o = make_float4(rstd * (d.x * ga.x - c1 - xh.x * c2), rstd * (d.y * ga.y - c1 - xh.y * c2),
                            rstd * (d.z * ga.z - c1 - xh.z * c2), rstd * (d.w * ga.w - c1 - xh.w * c2));
        }
        st4(p.y, p.y_f32, idx, o);
    }
}

static int gn_fill(GnP& p, const void* x, int x_dtype, const float* gamma, const float* beta, float* stats, void* scratch, int B,
                   int HW, int C, int G) {
    if (!(x && gamma && beta && stats && scratch && B > 0 && HW > 0 && G > 0)) return UENC_EINVAL;
    if (!(C % G == 0 && (C / G) % 4 == 0 && 64 % (C / G) == 0 && C <= 1024 && GN_THREADS % (C / 4) == 0)) return UENC_EINVAL;
    if (!(x_dtype == UENC_F32 || x_dtype == UENC_BF16) || ((uintptr_t)x & 15) || ((uintptr_t)scratch & 15)) return UENC_EINVAL;
    p.x = x; p.x_f32 = (x_dtype == UENC_F32); p.gamma = gamma; p.beta = beta; p.stats = stats;
    p.part = (double*)scratch;
    p.B = B; p.HW = HW; p.C = C; p.G = G;
    int nchunk = (HW + 127) / 128;                    // >= 4 workgroups per CU at 1/4 resolution; 128+ tokens each
    if (nchunk > 1024) nchunk = 1024;
    p.nchunk = nchunk;
    p.coef = nullptr; p.dgamma = p.dbeta = nullptr; p.add_src = nullptr; p.dy = nullptr; p.y = nullptr; p.relu = 0;
    p.H = p.W = p.Hs = p.Ws = 0; p.eps = 0.f; p.y_f32 = 0; p.dy_f32 = 0;
    return UENC_OK;
}

// scratch bytes for uenc_groupnorm_tokens_fwd / _bwd: fp64 partial sums (B, nchunk <= 1024, C, 2) + (B, G, 2) floats
extern "C" long uenc_groupnorm_tokens_scratch_bytes(int B, int HW, int C, int G) {
    long nchunk = ((long)HW + 127) / 128;
    if (nchunk > 1024) nchunk = 1024;
    return (long)B * nchunk * C * 2 * 8 + (long)B * G * 2 * 4 + 256;
}

// y = GroupNorm(x) [+ bilinear_resize(add_src)] [ReLU] on token matrices; stats (B, G, 2) = (mean, rstd) is written for
// the backward.  add_src: optional fp32 (B, Hs, Ws, C) map (then HW == H * W).
extern "C" int uenc_groupnorm_tokens_fwd(const void* x, int x_dtype, const float* gamma, const float* beta, void* y, int y_dtype,
                                         float* stats, void* scratch, const float* add_src, int Hs, int Ws, int H, int W, int B, int HW,
                                         int C, int G, float eps, int relu, hipStream_t stream) {
    GnP p;
    int rc = gn_fill(p, x, x_dtype, gamma, beta, stats, scratch, B, HW, C, G);
    if (rc != UENC_OK) return rc;
    UENC_CHECK_ARG(y && (y_dtype == UENC_F32 || y_dtype == UENC_BF16) && ((uintptr_t)y & 7) == 0);
    UENC_CHECK_ARG(add_src == nullptr || (Hs > 0 && Ws > 0 && H > 0 && W > 0 && (long)H * W == HW));
    p.y = y; p.y_f32 = (y_dtype == UENC_F32); p.eps = eps; p.relu = relu; p.add_src = add_src; p.H = H; p.W = W; p.Hs = Hs; p.Ws = Ws;
    hipLaunchKernelGGL(gn_reduce_kernel<0>, dim3(p.nchunk, B), dim3(GN_THREADS), 0, stream, p);
    hipLaunchKernelGGL(gn_stats_kernel, dim3(B * G), dim3(64), 0, stream, p);
    const long total = (long)B * HW * (C / 4);
    long blocks = (total + GN_THREADS - 1) / GN_THREADS;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(gn_apply_kernel<false>, dim3((unsigned)blocks), dim3(GN_THREADS), 0, stream, p);
    UENC_LAUNCH_RET();
}

// dx = dGroupNorm (with the ReLU mask recomputed from x when relu != 0); dgamma / dbeta (C) are accumulated (+=).
extern "C" int uenc_groupnorm_tokens_bwd(const void* dy, int dy_dtype, const void* x, int x_dtype, const float* gamma, const float* beta,
                                         const float* stats, void* dx, int dx_dtype, float* dgamma, float* dbeta, void* scratch, int B,
                                         int HW, int C, int G, int relu, hipStream_t stream) {
    GnP p;
    int rc = gn_fill(p, x, x_dtype, gamma, beta, (float*)stats, scratch, B, HW, C, G);
    if (rc != UENC_OK) return rc;
    UENC_CHECK_ARG(dy && dx && (dy_dtype == UENC_F32 || dy_dtype == UENC_BF16) && (dx_dtype == UENC_F32 || dx_dtype == UENC_BF16));
    UENC_CHECK_ARG((((uintptr_t)dy | (uintptr_t)dx) & 7) == 0);
    p.dy = dy; p.dy_f32 = (dy_dtype == UENC_F32); p.y = dx; p.y_f32 = (dx_dtype == UENC_F32); p.relu = relu;
    p.dgamma = dgamma; p.dbeta = dbeta;
    p.coef = (float*)((char*)scratch + (long)B * p.nchunk * C * 2 * 8);
    hipLaunchKernelGGL(gn_reduce_kernel<1>, dim3(p.nchunk, B), dim3(GN_THREADS), 0, stream, p);
    hipLaunchKernelGGL(gn_bwd_coef_kernel, dim3(B * ((C + 63) / 64)), dim3(GN_THREADS), 0, stream, p);
    const long total = (long)B * HW * (C / 4);
    long blocks = (total + GN_THREADS - 1) / GN_THREADS;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(gn_apply_kernel<true>, dim3((unsigned)blocks), dim3(GN_THREADS), 0, stream, p);
    UENC_LAUNCH_RET();
}

// ---- adjoint of the bilinear merge: dsrc[b, i, j, :] = sum over the output pixels whose taps include (i, j) ----------------
// Gather form (no atomics): for a source index i the candidate outputs are those with source coordinate in (i - 1, i + 1);
// each candidate's forward taps are recomputed exactly as in bilinear4, so clamping at the borders is mirrored bit for bit.
struct UpB { const void* dy; int dy_f32; float* dsrc; int B, H, W, Hs, Ws, C; };

// weight with which output index o taps source index i along one axis (0 if it does not), exactly as the forward computes it
__device__ __forceinline__ float axis_weight(int o, int i, int n_in, float scale) {
    float f = ((float)o + 0.5f) * scale - 0.5f;
    f = f < 0.f ? 0.f : f;
    const int i0 = min((int)f, n_in - 1), i1 = min(i0 + 1, n_in - 1);
    const float l = f - (float)i0;
    return (i0 == i ? 1.f - l : 0.f) + (i1 == i ? l : 0.f);
}
// candidate outputs for source index i: source coordinate within (i - 1, i + 1), one extra on either side for rounding
__device__ __forceinline__ void axis_range(int i, int n_out, float scale, int& lo, int& hi) {
    const float inv = 1.0f / scale;
    lo = (int)floorf(((float)i - 0.5f) * inv - 0.5f) - 1;
    hi = (int)ceilf(((float)i + 1.5f) * inv - 0.5f) + 1;
    if (i == 0) lo = 0;                         // every output clamped to coordinate 0 taps index 0
    lo = lo < 0 ? 0 : lo;
    hi = hi > n_out - 1 ? n_out - 1 : hi;
}

__global__ __launch_bounds__(256) void upsample_bwd_tokens_kernel(UpB p) {
    const int cqn = p.C >> 2;
    const long total = (long)p.B * p.Hs * p.Ws * cqn;
    const float sy = (float)p.Hs / (float)p.H, sx = (float)p.Ws / (float)p.W;
    for (long t = (long)blockIdx.x * 256 + threadIdx.x; t < total; t += (long)gridDim.x * 256) {
        const int cq = (int)(t % cqn);
        long r = t / cqn;
        const int j = (int)(r % p.Ws); r /= p.Ws;
        const int i = (int)(r % p.Hs);
        const int b = (int)(r / p.Hs);
        int ylo, yhi, xlo, xhi;
        axis_range(i, p.H, sy, ylo, yhi);
        axis_range(j, p.W, sx, xlo, xhi);
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int oy = ylo; oy <= yhi; ++oy) {
            const float wy = axis_weight(oy, i, p.Hs, sy);
            if (wy == 0.f) continue;
            for (int ox = xlo; ox <= xhi; ++ox) {
                const float w = wy * axis_weight(ox, j, p.Ws, sx);
                if (w == 0.f) continue;
                const float4 d = ld4(p.dy, p.dy_f32, (((long)b * p.H + oy) * p.W + ox) * p.C + cq * 4);
                acc.x += w * d.x; acc.y += w * d.y; acc.z += w * d.z; acc.w += w * d.w;
            }
        }
        *(float4*)(p.dsrc + (((long)b * p.Hs + i) * p.Ws + j) * p.C + cq * 4) = acc;
    }
}

// dsrc (B, Hs, Ws, C) fp32 (overwritten) = adjoint of bilinear_resize (Hs, Ws) -> (H, W) applied to dy (B, H, W, C), H >= Hs, W >= Ws.
extern "C" int uenc_upsample_bilinear_tokens_bwd(const void* dy, int dy_dtype, float* dsrc, int B, int H, int W, int Hs, int Ws, int C,
                                                 hipStream_t stream) {
    UENC_CHECK_ARG(dy && dsrc && B > 0 && H > 0 && W > 0 && Hs > 0 && Ws > 0 && C > 0 && C % 4 == 0);
    UENC_CHECK_ARG((dy_dtype == UENC_F32 || dy_dtype == UENC_BF16) && H >= Hs && W >= Ws);
    UpB p{dy, dy_dtype == UENC_F32, dsrc, B, H, W, Hs, Ws, C};
    const long total = (long)B * Hs * Ws * (C / 4);
    long blocks = (total + 255) / 256;
    if (blocks > 16384) blocks = 16384;
    hipLaunchKernelGGL(upsample_bwd_tokens_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, p);
    UENC_LAUNCH_RET();
}

// ---- 3x3, stride 1, pad 1 convolution as a GEMM: patch matrix and its adjoint (bf16, 16 bytes = 8 channels per thread) ----
// col[(b, y, x)][(ky, kx, c)] = in[b, y + ky - 1, x + kx - 1, c] (0 outside)
__global__ __launch_bounds__(256) void im2col3x3_kernel(const bf16* __restrict__ in, bf16* __restrict__ col, int B, int H, int W, int C) {
    const int c8n = C >> 3;
    const long total = (long)B * H * W * 9 * c8n;
    for (long t = (long)blockIdx.x * 256 + threadIdx.x; t < total; t += (long)gridDim.x * 256) {
        const int c8 = (int)(t % c8n);
        long r = t / c8n;
        const int k = (int)(r % 9); r /= 9;
        const int x = (int)(r % W); r /= W;
        const int y = (int)(r % H);
        const int b = (int)(r / H);
        const int yy = y + k / 3 - 1, xx = x + k % 3 - 1;
        u32x4 v = {0u, 0u, 0u, 0u};
        if (yy >= 0 && yy < H && xx >= 0 && xx < W) v = *(const u32x4*)(in + (((long)b * H + yy) * W + xx) * C + c8 * 8);
        *(u32x4*)(col + t * 8) = v;
    }
}

// dx[b, y, x, c] = sum_{ky, kx} dcol[(b, y - ky + 1, x - kx + 1)][(ky, kx, c)]
__global__ __launch_bounds__(256) void col2im3x3_kernel(const bf16* __restrict__ dcol, bf16* __restrict__ dx, int B, int H, int W, int C) {
    const int c8n = C >> 3;
    const long total = (long)B * H * W * c8n;
    for (long t = (long)blockIdx.x * 256 + threadIdx.x; t < total; t += (long)gridDim.x * 256) {
        const int c8 = (int)(t % c8n);
        long r = t / c8n;
        const int x = (int)(r % W); r /= W;
        const int y = (int)(r % H);
        const int b = (int)(r / H);
        float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int k = 0; k < 9; ++k) {
            const int yy = y - (k / 3) + 1, xx = x - (k % 3) + 1;
            if (yy >= 0 && yy < H && xx >= 0 && xx < W) {
                const bf16x8 v = *(const bf16x8*)(dcol + ((((long)b * H + yy) * W + xx) * 9 + k) * C + c8 * 8);
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[j] += (float)v[j];
            }
        }
        bf16x8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = (bf16)acc[j];
        *(bf16x8*)(dx + t * 8) = o;
    }
}

extern "C" int uenc_im2col3x3(const void* in, void* col, int B, int H, int W, int C, hipStream_t stream) {
    UENC_CHECK_ARG(in && col && B > 0 && H > 0 && W > 0 && C > 0 && C % 8 == 0 && (((uintptr_t)in | (uintptr_t)col) & 15) == 0);
    const long total = (long)B * H * W * 9 * (C / 8);
    long blocks = (total + 255) / 256;
    if (blocks > 65536) blocks = 65536;
    hipLaunchKernelGGL(im2col3x3_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, (const bf16*)in, (bf16*)col, B, H, W, C);
    UENC_LAUNCH_RET();
}

extern "C" int uenc_col2im3x3(const void* dcol, void* dx, int B, int H, int W, int C, hipStream_t stream) {
    UENC_CHECK_ARG(dcol && dx && B > 0 && H > 0 && W > 0 && C > 0 && C % 8 == 0 && (((uintptr_t)dcol | (uintptr_t)dx) & 15) == 0);
    const long total = (long)B * H * W * (C / 8);
    long blocks = (total + 255) / 256;
    if (blocks > 65536) blocks = 65536;
    hipLaunchKernelGGL(col2im3x3_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, (const bf16*)dcol, (bf16*)dx, B, H, W, C);
    UENC_LAUNCH_RET();
}

// ---- 3x3, stride 2, pad 1 convolution as a GEMM (DiNAT ConvTokenizer / ConvDownsampler, reference backbone/dinat.py:17-45) ----
// col[(b, yo, xo)][(ky, kx, c)] = in[b, 2 yo + ky - 1, 2 xo + kx - 1, c] (0 outside); Ho = ceil(H / 2), Wo = ceil(W / 2)
__global__ __launch_bounds__(256) void im2col3x3_s2_kernel(const bf16* __restrict__ in, bf16* __restrict__ col, int B, int H, int W, int C,
                                                           int Ho, int Wo) {
    const int c8n = C >> 3;
    const long total = (long)B * Ho * Wo * 9 * c8n;
    for (long t = (long)blockIdx.x * 256 + threadIdx.x; t < total; t += (long)gridDim.x * 256) {
        const int c8 = (int)(t % c8n);
        long r = t / c8n;
        const int k = (int)(r % 9); r /= 9;
        const int xo = (int)(r % Wo); r /= Wo;
        const int yo = (int)(r % Ho);
        const int b = (int)(r / Ho);
        const int yy = 2 * yo + k / 3 - 1, xx = 2 * xo + k % 3 - 1;
        u32x4 v = {0u, 0u, 0u, 0u};
        if (yy >= 0 && yy < H && xx >= 0 && xx < W) v = *(const u32x4*)(in + (((long)b * H + yy) * W + xx) * C + c8 * 8);
        *(u32x4*)(col + t * 8) = v;
    }
}

// dx[b, y, x, c] (fp32) = sum over the taps (ky, kx) with y + 1 - ky and x + 1 - kx even and the output pixel in range of
// dcol[(b, (y + 1 - ky) / 2, (x + 1 - kx) / 2)][(ky, kx, c)]: a gather, every dx element written once
__global__ __launch_bounds__(256) void col2im3x3_s2_kernel(const bf16* __restrict__ dcol, float* __restrict__ dx, int B, int H, int W, int C,
                                                           int Ho, int Wo) {
    const int c8n = C >> 3;
    const long total = (long)B * H * W * c8n;
    for (long t = (long)blockIdx.x * 256 + threadIdx.x; t < total; t += (long)gridDim.x * 256) {
        const int c8 = (int)(t % c8n);
        long r = t / c8n;
        const int x = (int)(r % W); r /= W;
        const int y = (int)(r % H);
        const int b = (int)(r / H);
        float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            const int ty = y + 1 - ky;
            if (ty < 0 || (ty & 1) || (ty >> 1) >= Ho) continue;
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const int tx = x + 1 - kx;
                if (tx < 0 || (tx & 1) || (tx >> 1) >= Wo) continue;
                const bf16x8 v = *(const bf16x8*)(dcol + ((((long)b * Ho + (ty >> 1)) * Wo + (tx >> 1)) * 9 + ky * 3 + kx) * C + c8 * 8);
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[j] += (float)v[j];
            }
        }
        float* o = dx + t * 8;
        *(float4*)o = make_float4(acc[0], acc[1], acc[2], acc[3]);
        *(float4*)(o + 4) = make_float4(acc[4], acc[5], acc[6], acc[7]);
    }
}

extern "C" int uenc_im2col3x3_s2(const void* in, void* col, int B, int H, int W, int C, hipStream_t stream) {
    UENC_CHECK_ARG(in && col && B > 0 && H > 0 && W > 0 && C > 0 && C % 8 == 0 && (((uintptr_t)in | (uintptr_t)col) & 15) == 0);
    const int Ho = (H + 1) / 2, Wo = (W + 1) / 2;
    const long total = (long)B * Ho * Wo * 9 * (C / 8);
    long blocks = (total + 255) / 256;
    if (blocks > 65536) blocks = 65536;
    hipLaunchKernelGGL(im2col3x3_s2_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, (const bf16*)in, (bf16*)col, B, H, W, C, Ho, Wo);
    UENC_LAUNCH_RET();
}

extern "C" int uenc_col2im3x3_s2(const void* dcol, float* dx, int B, int H, int W, int C, hipStream_t stream) {
    UENC_CHECK_ARG(dcol && dx && B > 0 && H > 0 && W > 0 && C > 0 && C % 8 == 0 && (((uintptr_t)dcol | (uintptr_t)dx) & 15) == 0);
    const int Ho = (H + 1) / 2, Wo = (W + 1) / 2;
    const long total = (long)B * H * W * (C / 8);
    long blocks = (total + 255) / 256;
    if (blocks > 65536) blocks = 65536;
    hipLaunchKernelGGL(col2im3x3_s2_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, (const bf16*)dcol, dx, B, H, W, C, Ho, Wo);
    UENC_LAUNCH_RET();
}

