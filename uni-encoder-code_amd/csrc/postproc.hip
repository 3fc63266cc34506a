// Segmentation post-processing on the device, fused with the mask upsample so that the (Q, H, W) fp32
// mask tensor -- 1.25 GB per 1024 x 2048 image at Q = 150, written by F.interpolate and read back twice by the reference
// (model/oneformer_model.py:255-263, then semantic_inference :367-371 / panoptic_inference :373-434) -- is never materialised:
// every kernel reads the low-resolution mask logits (Q, h, w) (L2-resident, 78 MB per image) and interpolates on the fly.
//
//   uenc_postproc_semantic        sem[c, y, x] = sum_q P[q, c] * sigmoid(up(m_q)[y, x])                       (:367-371)
//   uenc_postproc_panoptic_stats  ids[y, x] = argmax_q score_q * sigmoid(up(m_q)[y, x]) and, per query, the three pixel counts
//                                 the reference takes with 3 Q `.item()` syncs: |ids == q|, |sigmoid >= 0.5|, |both|  (:399-408)
//   uenc_postproc_panoptic_label  panoptic_seg[y, x] = segment_id[ids] where that query's sigmoid >= 0.5, else 0     (:420-425)
//
// `up` is F.interpolate(mode="bilinear", align_corners=False) to the padded input size (:258-263) in PyTorch's operation order;
// cropping the padding away (detectron2's sem_seg_postprocess) is the output extent (Ho, Wo) <= (Hp, Wp).  The final resize to a
// different output resolution is not fused (the caller falls back to separate passes then).
// Bound: VALU -- Q x (4 taps + sigmoid + C FMAs) per output pixel; HBM traffic is the C (or 1) output planes.
#include "common.h"

struct UpGeom { int hl, wl, Ho, Wo; float sy, sx; };

struct Tap { int o00, o01, o10, o11; float hy, ly, hx, lx; };

__device__ __forceinline__ Tap make_tap(const UpGeom& g, int y, int x) {
    Tap t;
    float fy = ((float)y + 0.5f) * g.sy - 0.5f;
    fy = fy < 0.f ? 0.f : fy;
    const int y0 = min((int)fy, g.hl - 1), y1 = min(y0 + 1, g.hl - 1);
    t.ly = fy - (float)y0; t.hy = 1.f - t.ly;
    float fx = ((float)x + 0.5f) * g.sx - 0.5f;
    fx = fx < 0.f ? 0.f : fx;
    const int x0 = min((int)fx, g.wl - 1), x1 = min(x0 + 1, g.wl - 1);
    t.lx = fx - (float)x0; t.hx = 1.f - t.lx;
    t.o00 = y0 * g.wl + x0; t.o01 = y0 * g.wl + x1; t.o10 = y1 * g.wl + x0; t.o11 = y1 * g.wl + x1;
    return t;
}

__device__ __forceinline__ float tap_value(const float* __restrict__ m, const Tap& t) {
    return t.hy * (t.hx * m[t.o00] + t.lx * m[t.o01]) + t.ly * (t.hx * m[t.o10] + t.lx * m[t.o11]);
}

__device__ __forceinline__ float sigmoid_f(float v) { return 1.0f / (1.0f + expf(-v)); }

// grid (ceil(Wo / 256), Ho, class tiles of 32); P is (Q, Cp) fp32 with Cp a multiple of 32 (zero padded)
__global__ __launch_bounds__(256) void postproc_semantic_kernel(const float* __restrict__ ml, const float* __restrict__ P, float* __restrict__ sem,
                                                                 int Q, int C, int Cp, UpGeom g) {
    const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y, c0 = blockIdx.z * 32;
    if (x >= g.Wo) return;
    const Tap t = make_tap(g, y, x);
    const long plane = (long)g.hl * g.wl;
    float acc[32];
#pragma unroll
    for (int c = 0; c < 32; ++c) acc[c] = 0.f;
    for (int q = 0; q < Q; ++q) {
        const float s = sigmoid_f(tap_value(ml + q * plane, t));
        const float* pq = P + (long)q * Cp + c0;             // wave-uniform: scalar loads
#pragma unroll
        for (int c = 0; c < 32; ++c) acc[c] += pq[c] * s;
    }
    const int nc = min(32, C - c0);
    for (int c = 0; c < nc; ++c) sem[((long)(c0 + c) * g.Ho + y) * g.Wo + x] = acc[c];
}

// score (Q): softmax score of kept queries, 0 for the others (a kept query's product is > 0, so the others never win; with no
// kept query at all every segment id is 0 anyway).  counts (3, Q) int32, zeroed by the caller.  grid (ceil(Wo / 256), Ho)
__global__ __launch_bounds__(256) void postproc_panoptic_stats_kernel(const float* __restrict__ ml, const float* __restrict__ score,
                                                                       int* __restrict__ ids, int* __restrict__ counts, int Q, UpGeom g) {
    extern __shared__ int cnt[];                              // [3][Q]
    for (int i = threadIdx.x; i < 3 * Q; i += 256) cnt[i] = 0;
    __syncthreads();
    const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
    if (x < g.Wo) {
        const Tap t = make_tap(g, y, x);
        const long plane = (long)g.hl * g.wl;
        float best = -1.f, bestp = 0.f;
        int bi = 0;
        for (int q = 0; q < Q; ++q) {
            const float sc = score[q];                       // uniform
            if (sc <= 0.f) continue;
            const float pr = sigmoid_f(tap_value(ml + q * plane, t));
            const unsigned long long over = __ballot(pr >= 0.5f);        // one LDS atomic per wave and query, not one per pixel
            if (over != 0ull && (threadIdx.x & 63) == 0) atomicAdd(&cnt[Q + q], (int)__popcll(over));
            const float v = sc * pr;
            if (v > best) { best = v; bi = q; bestp = pr; }  // strict: the first maximum wins, as torch.argmax
        }
        ids[(long)y * g.Wo + x] = bi;
        if (best > 0.f) {
            atomicAdd(&cnt[bi], 1);
            if (bestp >= 0.5f) atomicAdd(&cnt[2 * Q + bi], 1);
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 3 * Q; i += 256)
        if (cnt[i]) atomicAdd(counts + i, cnt[i]);
}

// segid (Q) int32: the segment id of every query (0 = dropped / not kept)
__global__ __launch_bounds__(256) void postproc_panoptic_label_kernel(const float* __restrict__ ml, const int* __restrict__ ids,
                                                                       const int* __restrict__ segid, int* __restrict__ seg, UpGeom g) {
    const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
    if (x >= g.Wo) return;
    const int q = ids[(long)y * g.Wo + x];
    const int sid = segid[q];
    int out = 0;
    if (sid != 0) {
        const Tap t = make_tap(g, y, x);
        if (sigmoid_f(tap_value(ml + (long)q * g.hl * g.wl, t)) >= 0.5f) out = sid;
    }
    seg[(long)y * g.Wo + x] = out;
}

static int geom(UpGeom& g, int hl, int wl, int Hp, int Wp, int Ho, int Wo) {
    UENC_CHECK_ARG(hl > 0 && wl > 0 && Hp > 0 && Wp > 0 && Ho > 0 && Wo > 0 && Ho <= Hp && Wo <= Wp && Ho <= 65535);
    g.hl = hl; g.wl = wl; g.Ho = Ho; g.Wo = Wo; g.sy = (float)hl / (float)Hp; g.sx = (float)wl / (float)Wp;
    return UENC_OK;
}

extern "C" int uenc_postproc_semantic(const float* mask_logits, const float* class_prob, float* sem, int Q, int C, int Cp, int hl, int wl,
                                      int Hp, int Wp, int Ho, int Wo, hipStream_t stream) {
    UENC_CHECK_ARG(mask_logits && class_prob && sem && Q > 0 && C > 0 && Cp >= C && Cp % 32 == 0);
    UpGeom g;
    const int rc = geom(g, hl, wl, Hp, Wp, Ho, Wo);
    if (rc != UENC_OK) return rc;
    hipLaunchKernelGGL(postproc_semantic_kernel, dim3((Wo + 255) / 256, Ho, (C + 31) / 32), dim3(256), 0, stream, mask_logits, class_prob, sem,
                       Q, C, Cp, g);
    UENC_LAUNCH_RET();
}

extern "C" int uenc_postproc_panoptic_stats(const float* mask_logits, const float* score, int* ids, int* counts, int Q, int hl, int wl, int Hp,
                                            int Wp, int Ho, int Wo, hipStream_t stream) {
    UENC_CHECK_ARG(mask_logits && score && ids && counts && Q > 0 && Q <= 4096);
    UpGeom g;
    const int rc = geom(g, hl, wl, Hp, Wp, Ho, Wo);
    if (rc != UENC_OK) return rc;
    hipLaunchKernelGGL(postproc_panoptic_stats_kernel, dim3((Wo + 255) / 256, Ho), dim3(256), 3 * Q * sizeof(int), stream, mask_logits, score, ids,
                       counts, Q, g);
    UENC_LAUNCH_RET();
}

extern "C" int uenc_postproc_panoptic_label(const float* mask_logits, const int* ids, const int* segid, int* seg, int Q, int hl, int wl, int Hp,
                                            int Wp, int Ho, int Wo, hipStream_t stream) {
    UENC_CHECK_ARG(mask_logits && ids && segid && seg && Q > 0);
    UpGeom g;
    const int rc = geom(g, hl, wl, Hp, Wp, Ho, Wo);
    if (rc != UENC_OK) return rc;
    hipLaunchKernelGGL(postproc_panoptic_label_kernel, dim3((Wo + 255) / 256, Ho), dim3(256), 0, stream, mask_logits, ids, segid, seg, g);
    UENC_LAUNCH_RET();
}
