// Fused (shifted-)window attention of the Swin backbone, forward and backward (gfx950).
//
// Replaces, in one kernel each way, the reference's  F.pad -> torch.roll -> window_partition ->
// [q*scale @ k^T + relative_position_bias (+ shift mask) -> softmax -> @ v] -> window_reverse ->
// torch.roll -> crop   (model/modeling/backbone/swin.py:250-289 around WindowAttention.forward
// :131-171, mask built at :414-440).  Pad / roll / partition / reverse / crop are pure addressing
// here: nothing is copied, the N x N score matrix never leaves the CU.
//
// Data layout in HBM: qkv is the qkv Linear's output for the H*W real tokens, token-major
// (B, H, W, 3C) bf16, q | k | v each C wide, head h at [32h, 32h+32).  Padding slots (the zero rows
// the reference pads after norm1, swin.py:254) never exist in memory: their q/k/v are the qkv bias
// (Linear of a zero row), read from `qkv_bias`; they take part as un-masked keys exactly as in the
// reference and their outputs are dropped.
//
// One workgroup per (window, head): WAVES waves, each owning QT 16-query tiles.
//   S^T[key][q] = K Q^T        MFMA 16x16x32 (head_dim 32 = one k-step), operands straight from L2
//   softmax over keys          in registers: a lane holds one query column, 4 keys per tile
//   O^T[d][q]   = V^T P^T      P^T tiles feed the next MFMA as B operand with no lane movement;
//                              V^T comes from an LDS image [d][key] filled once per workgroup
// Backward recomputes S in both orientations (key-major for dQ, query-major for dK / dV) so every
// gradient of a window is produced by its own workgroup: no cross-workgroup reduction except the
// relative-position table (LDS accumulation, then one contiguous atomic burst per workgroup) and
// the qkv-bias gradient that padding slots contribute.
// Algorithmic HBM bytes per token per head-slice: read 3*32*2 (fwd) ; bwd read 5*32*2 + write 3*32*2.
#include "common.h"

struct WAttn {
    const bf16* qkv;        // (B, H, W, 3C)
    const bf16* qkv_bias;   // (3C) bf16 copy of attn.qkv.bias
    const float* bias_q;    // (nH, NP, NP) [h][q][key]   expanded relative-position bias, -30000 for key >= N
    const float* bias_k;    // (nH, NP, NP) [h][key][q]   same, key-major (backward phase B)
    bf16* out;              // (B, H, W, C) attention output (before proj)
    // backward only
    const bf16* o_saved;    // forward output
    const bf16* d_out;      // (B, H, W, C)
    bf16* dqkv;             // (B, H, W, 3C)
    float* dtab;            // (nH, (2ws-1)^2) accumulated
    float* dbias_pad;       // (3C) accumulated: gradient reaching qkv.bias through padding slots
    int B, H, W, C, nH, ws, shift, Hp, Wp, nWw, nWin, nWinTotal, N;
    float scale;
};

template <int NTILES>
struct WCfg {
    static constexpr int WAVES = (NTILES + 2) / 3;
    static constexpr int QT = (NTILES + WAVES - 1) / WAVES;
    static constexpr int NP = NTILES * 16;
    static constexpr int NKB = (NTILES + 1) / 2;       // 32-key blocks
    static constexpr int KP = NKB * 32 > 128 ? 264 : 136;   // LDS row pitch (elements): 16 B mod 256 B
};

__device__ __forceinline__ int region3(int v, int P, int ws, int shift) { return (v >= P - ws) + (v >= P - shift); }

// token bookkeeping shared by forward and backward: for slot t of this window,
//   tokoff = flat token index (b*H + h)*W + w, or -1 for a padding slot, or -2 beyond N
template <int NP>
__device__ __forceinline__ void window_slots(const WAttn& p, int b, int wi, int wj, int* tokoff, unsigned char* rid,
                                             unsigned short* yx, int nthreads) {
    for (int t = threadIdx.x; t < NP; t += nthreads) {
        int off = -2, r = 0, code = 0;
        if (t < p.N) {
            const int ty = t / p.ws, tx = t - ty * p.ws;
            const int hs = wi * p.ws + ty, wx = wj * p.ws + tx;
            int ho = hs + p.shift, wo = wx + p.shift;
            if (ho >= p.Hp) ho -= p.Hp;
            if (wo >= p.Wp) wo -= p.Wp;
            off = (ho < p.H && wo < p.W) ? (b * p.H + ho) * p.W + wo : -1;
            if (p.shift > 0) r = 3 * region3(hs, p.Hp, p.ws, p.shift) + region3(wx, p.Wp, p.ws, p.shift);
            code = (ty << 8) | tx;
        }
        tokoff[t] = off;
        rid[t] = (unsigned char)r;
        if (yx != nullptr) yx[t] = (unsigned short)code;
    }
}

__device__ __forceinline__ void decode_block(const WAttn& p, int& win, int& head) {
    // 8 consecutive block ids = 8 windows (one per XCD under round-robin dispatch), same head; the
    // heads of one window therefore share an XCD / L2 and the 128-byte lines they split.
    const int per = 8 * p.nH;
    const int group = blockIdx.x / per, r = blockIdx.x - group * per;
    win = group * 8 + (r & 7);
    head = r >> 3;
}

__device__ __forceinline__ bf16x8 zero8() {
    bf16x8 z;
#pragma unroll
    for (int i = 0; i < 8; ++i) z[i] = (bf16)0.f;
    return z;
}

// stage rows [key][32 d] (one head slice) of `base` transposed into LDS dst[d][key]
template <int KP, int NKEYS>
__device__ __forceinline__ void stage_transposed(bf16* dst, const bf16* base, long rowstride, const bf16* padrow,
                                                 bool pad_zero, int coloff, const int* tokoff, int nthreads) {
    for (int idx = threadIdx.x; idx < NKEYS * 4; idx += nthreads) {
        const int key = idx >> 2, ch = idx & 3;
        const int tok = tokoff[key];
        bf16x8 v = zero8();
        if (tok >= 0) v = *(const bf16x8*)(base + (long)tok * rowstride + coloff + ch * 8);
        else if (tok == -1 && !pad_zero) v = *(const bf16x8*)(padrow + coloff + ch * 8);
#pragma unroll
        for (int j = 0; j < 8; ++j) dst[(ch * 8 + j) * KP + key] = v[j];
    }
}

template <int NTILES>
__global__ __launch_bounds__(64 * WCfg<NTILES>::WAVES) void wattn_fwd_kernel(WAttn p) {
    using Cf = WCfg<NTILES>;
    constexpr int WAVES = Cf::WAVES, QT = Cf::QT, NP = Cf::NP, NKB = Cf::NKB, KP = Cf::KP, NK2 = NKB * 32;
    __shared__ __attribute__((aligned(16))) unsigned char smem[32 * KP * 2 + NK2 * 4 + NK2];
    bf16* Vt = (bf16*)smem;
    int* tokoff = (int*)(smem + 32 * KP * 2);
    unsigned char* rid = smem + 32 * KP * 2 + NK2 * 4;

    int win, head;
    decode_block(p, win, head);
    if (win >= p.nWinTotal) return;
    const int b = win / p.nWin, wrem = win - b * p.nWin;
    const int wi = wrem / p.nWw, wj = wrem - wi * p.nWw;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, fr = lane & 15, fg = lane >> 4;
    const int C = p.C, C3 = 3 * p.C, hoff = head * 32;

    window_slots<NP>(p, b, wi, wj, tokoff, rid, nullptr, 64 * WAVES);
    for (int t = NP + threadIdx.x; t < NK2; t += 64 * WAVES) { tokoff[t] = -2; rid[t] = 0; }
    __syncthreads();
    stage_transposed<KP, NK2>(Vt, p.qkv, C3, p.qkv_bias, false, 2 * C + hoff, tokoff, 64 * WAVES);
    __syncthreads();

    auto rowp = [&](int tok) -> const bf16* { return tok >= 0 ? p.qkv + (long)tok * C3 : p.qkv_bias; };

    bf16x8 qf[QT];
    int qtok[QT];
#pragma unroll
    for (int jq = 0; jq < QT; ++jq) {
        const int qt = wave * QT + jq;
        qtok[jq] = qt < NTILES ? tokoff[qt * 16 + fr] : -2;
        qf[jq] = *(const bf16x8*)(rowp(qtok[jq]) + hoff + 8 * fg);
    }
    f32x4 s[NTILES][QT];
#pragma unroll
    for (int kt = 0; kt < NTILES; ++kt) {
        const bf16x8 kf = *(const bf16x8*)(rowp(tokoff[kt * 16 + fr]) + C + hoff + 8 * fg);
#pragma unroll
        for (int jq = 0; jq < QT; ++jq) s[kt][jq] = mfma16(kf, qf[jq], (f32x4){0.f, 0.f, 0.f, 0.f});
    }
    float inv[QT];
#pragma unroll
    for (int jq = 0; jq < QT; ++jq) {
        const int qt = wave * QT + jq;
        const int qi = (qt < NTILES ? qt : 0) * 16 + fr;
        const int ridq = rid[qi];
        const float* brow = p.bias_q + ((long)head * NP + qi) * NP;
        float mx = -1e30f;
#pragma unroll
        for (int kt = 0; kt < NTILES; ++kt) {
            const float4 bb = *(const float4*)(brow + kt * 16 + 4 * fg);
            const unsigned rk = *(const unsigned*)(rid + kt * 16 + 4 * fg);
            const float bv[4] = {bb.x, bb.y, bb.z, bb.w};
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float v = s[kt][jq][r] * p.scale + bv[r];
                if (p.shift > 0 && (int)((rk >> (8 * r)) & 0xffu) != ridq) v -= 100.0f;
                s[kt][jq][r] = v;
                mx = fmaxf(mx, v);
            }
        }
        mx = fmaxf(mx, __shfl_xor(mx, 16));
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        float sum = 0.f;
#pragma unroll
        for (int kt = 0; kt < NTILES; ++kt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float e = __expf(s[kt][jq][r] - mx);
                s[kt][jq][r] = e;
                sum += e;
            }
        sum += __shfl_xor(sum, 16);
        sum += __shfl_xor(sum, 32);
        inv[jq] = 1.0f / sum;
    }
    f32x4 o[2][QT];
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int jq = 0; jq < QT; ++jq) o[dt][jq] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb) {
        bf16x8 pb[QT];
#pragma unroll
        for (int jq = 0; jq < QT; ++jq) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                pb[jq][r] = (bf16)s[2 * kb][jq][r];
                pb[jq][4 + r] = (2 * kb + 1 < NTILES) ? (bf16)s[(2 * kb + 1 < NTILES) ? 2 * kb + 1 : 0][jq][r] : (bf16)0.f;
            }
        }
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
            const bf16* vrow = Vt + (dt * 16 + fr) * KP + 32 * kb + 4 * fg;
            const bf16x4 lo = *(const bf16x4*)vrow, hi = *(const bf16x4*)(vrow + 16);
            bf16x8 vf;
#pragma unroll
            for (int r = 0; r < 4; ++r) { vf[r] = lo[r]; vf[4 + r] = hi[r]; }
#pragma unroll
            for (int jq = 0; jq < QT; ++jq) o[dt][jq] = mfma16(vf, pb[jq], o[dt][jq]);
        }
    }
#pragma unroll
    for (int jq = 0; jq < QT; ++jq) {
        if (qtok[jq] < 0) continue;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
            bf16x4 ov;
#pragma unroll
            for (int r = 0; r < 4; ++r) ov[r] = (bf16)(o[dt][jq][r] * inv[jq]);
            *(bf16x4*)(p.out + (long)qtok[jq] * C + hoff + dt * 16 + 4 * fg) = ov;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// backward
// ------------------------------------------------------------------------------------------------
template <int NTILES>
__global__ __launch_bounds__(64 * WCfg<NTILES>::WAVES) void wattn_bwd_kernel(WAttn p) {
    using Cf = WCfg<NTILES>;
    constexpr int WAVES = Cf::WAVES, QT = Cf::QT, NP = Cf::NP, NKB = Cf::NKB, KP = Cf::KP, NK2 = NKB * 32;
    constexpr int NTH = 64 * WAVES;
    constexpr int OFF_TOK = 3 * 32 * KP * 2;
    constexpr int OFF_LSE = OFF_TOK + NK2 * 4;
    constexpr int OFF_DEL = OFF_LSE + NK2 * 4;
    constexpr int OFF_YX = OFF_DEL + NK2 * 4;
    constexpr int OFF_RID = OFF_YX + NK2 * 2;
    constexpr int OFF_TAB = (OFF_RID + NK2 + 15) / 16 * 16;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    bf16* Kt = (bf16*)smem;
    bf16* Qt = Kt + 32 * KP;
    bf16* dOt = Qt + 32 * KP;
    int* tokoff = (int*)(smem + OFF_TOK);
    float* lse = (float*)(smem + OFF_LSE);
    float* delta = (float*)(smem + OFF_DEL);
    unsigned short* yx = (unsigned short*)(smem + OFF_YX);
    unsigned char* rid = smem + OFF_RID;
    float* tab = (float*)(smem + OFF_TAB);
    const int T1 = 2 * p.ws - 1, TT = T1 * T1;

    int win, head;
    decode_block(p, win, head);
    if (win >= p.nWinTotal) return;
    const int b = win / p.nWin, wrem = win - b * p.nWin;
    const int wi = wrem / p.nWw, wj = wrem - wi * p.nWw;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, fr = lane & 15, fg = lane >> 4;
    const int C = p.C, C3 = 3 * p.C, hoff = head * 32;

    window_slots<NP>(p, b, wi, wj, tokoff, rid, yx, NTH);
    for (int t = NP + threadIdx.x; t < NK2; t += NTH) { tokoff[t] = -2; rid[t] = 0; yx[t] = 0; }
    for (int t = threadIdx.x; t < TT; t += NTH) tab[t] = 0.f;
    for (int t = threadIdx.x; t < NK2; t += NTH) { lse[t] = 0.f; delta[t] = 0.f; }
    __syncthreads();
    stage_transposed<KP, NK2>(Kt, p.qkv, C3, p.qkv_bias, false, C + hoff, tokoff, NTH);
    stage_transposed<KP, NK2>(Qt, p.qkv, C3, p.qkv_bias, false, hoff, tokoff, NTH);
    stage_transposed<KP, NK2>(dOt, p.d_out, C, nullptr, true, hoff, tokoff, NTH);
    __syncthreads();

    auto rowp = [&](int tok) -> const bf16* { return tok >= 0 ? p.qkv + (long)tok * C3 : p.qkv_bias; };
    auto load_do = [&](int tok) -> bf16x8 {
        return tok >= 0 ? *(const bf16x8*)(p.d_out + (long)tok * C + hoff + 8 * fg) : zero8();
    };
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};

    // ---------------- phase A: key-major scores, dQ, softmax statistics, dTable ----------------
    {
        bf16x8 qf[QT], dof[QT];
        int qtok[QT];
        float dl[QT];
#pragma unroll
        for (int jq = 0; jq < QT; ++jq) {
            const int qt = wave * QT + jq;
            qtok[jq] = qt < NTILES ? tokoff[qt * 16 + fr] : -2;
            qf[jq] = *(const bf16x8*)(rowp(qtok[jq]) + hoff + 8 * fg);
            dof[jq] = load_do(qtok[jq]);
            // delta[q] = sum_d dO[q][d] * O[q][d]
            float part = 0.f;
            if (qtok[jq] >= 0) {
                const bf16x8 ov = *(const bf16x8*)(p.o_saved + (long)qtok[jq] * C + hoff + 8 * fg);
#pragma unroll
                for (int j = 0; j < 8; ++j) part += (float)ov[j] * (float)dof[jq][j];
            }
            part += __shfl_xor(part, 16);
            part += __shfl_xor(part, 32);
            dl[jq] = part;
        }
        f32x4 s[NTILES][QT];
#pragma unroll
        for (int kt = 0; kt < NTILES; ++kt) {
            const bf16x8 kf = *(const bf16x8*)(rowp(tokoff[kt * 16 + fr]) + C + hoff + 8 * fg);
#pragma unroll
            for (int jq = 0; jq < QT; ++jq) s[kt][jq] = mfma16(kf, qf[jq], zero4);
        }
#pragma unroll
        for (int jq = 0; jq < QT; ++jq) {
            const int qt = wave * QT + jq;
            const int qi = (qt < NTILES ? qt : 0) * 16 + fr;
            const int ridq = rid[qi];
            const float* brow = p.bias_q + ((long)head * NP + qi) * NP;
            float mx = -1e30f;
#pragma unroll
            for (int kt = 0; kt < NTILES; ++kt) {
                const float4 bb = *(const float4*)(brow + kt * 16 + 4 * fg);
                const unsigned rk = *(const unsigned*)(rid + kt * 16 + 4 * fg);
                const float bv[4] = {bb.x, bb.y, bb.z, bb.w};
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float v = s[kt][jq][r] * p.scale + bv[r];
                    if (p.shift > 0 && (int)((rk >> (8 * r)) & 0xffu) != ridq) v -= 100.0f;
                    s[kt][jq][r] = v;
                    mx = fmaxf(mx, v);
                }
            }
            mx = fmaxf(mx, __shfl_xor(mx, 16));
            mx = fmaxf(mx, __shfl_xor(mx, 32));
            float sum = 0.f;
#pragma unroll
            for (int kt = 0; kt < NTILES; ++kt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float e = __expf(s[kt][jq][r] - mx);
                    s[kt][jq][r] = e;
                    sum += e;
                }
            sum += __shfl_xor(sum, 16);
            sum += __shfl_xor(sum, 32);
            const float inv = 1.0f / sum;
            if (qt < NTILES && fg == 0) { lse[qi] = mx + __logf(sum); delta[qi] = dl[jq]; }
            // dS^T = P * (dP^T - delta), dP^T[key][q] = V[key] . dO[q]
            const int qcode = yx[qi];
            const int qy = qcode >> 8, qx = qcode & 0xff;
            const bool qreal = qt < NTILES && qi < p.N;
#pragma unroll
            for (int kt = 0; kt < NTILES; ++kt) {
                const bf16x8 vf = *(const bf16x8*)(rowp(tokoff[kt * 16 + fr]) + 2 * C + hoff + 8 * fg);
                const f32x4 dp = mfma16(vf, dof[jq], zero4);
                const uint2 kc = *(const uint2*)(yx + kt * 16 + 4 * fg);
                const unsigned kcs[4] = {kc.x & 0xffffu, kc.x >> 16, kc.y & 0xffffu, kc.y >> 16};
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float ds = s[kt][jq][r] * inv * (dp[r] - dl[jq]);
                    s[kt][jq][r] = ds;
                    const int key = kt * 16 + 4 * fg + r;
                    if (qreal && key < p.N) {
                        const int ky = kcs[r] >> 8, kx = kcs[r] & 0xff;
                        atomicAdd(&tab[(qy - ky + p.ws - 1) * T1 + (qx - kx + p.ws - 1)], ds);
                    }
                }
            }
        }
        // dQ^T[d][q] = scale * sum_key K^T[d][key] dS^T[key][q]
        f32x4 dq[2][QT];
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int jq = 0; jq < QT; ++jq) dq[dt][jq] = zero4;
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb) {
            bf16x8 pb[QT];
#pragma unroll
            for (int jq = 0; jq < QT; ++jq)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    pb[jq][r] = (bf16)s[2 * kb][jq][r];
                    pb[jq][4 + r] = (2 * kb + 1 < NTILES) ? (bf16)s[(2 * kb + 1 < NTILES) ? 2 * kb + 1 : 0][jq][r] : (bf16)0.f;
                }
#pragma unroll
            for (int dt = 0; dt < 2; ++dt) {
                const bf16* krow = Kt + (dt * 16 + fr) * KP + 32 * kb + 4 * fg;
                const bf16x4 lo = *(const bf16x4*)krow, hi = *(const bf16x4*)(krow + 16);
                bf16x8 kf;
#pragma unroll
                for (int r = 0; r < 4; ++r) { kf[r] = lo[r]; kf[4 + r] = hi[r]; }
#pragma unroll
                for (int jq = 0; jq < QT; ++jq) dq[dt][jq] = mfma16(kf, pb[jq], dq[dt][jq]);
            }
        }
#pragma unroll
        for (int jq = 0; jq < QT; ++jq) {
            const int qt = wave * QT + jq;
            if (qt >= NTILES || qtok[jq] == -2) continue;
#pragma unroll
            for (int dt = 0; dt < 2; ++dt) {
                const int col = hoff + dt * 16 + 4 * fg;
                if (qtok[jq] >= 0) {
                    bf16x4 ov;
#pragma unroll
                    for (int r = 0; r < 4; ++r) ov[r] = (bf16)(dq[dt][jq][r] * p.scale);
                    *(bf16x4*)(p.dqkv + (long)qtok[jq] * C3 + col) = ov;
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r) atomicAdd(p.dbias_pad + col + r, dq[dt][jq][r] * p.scale);
                }
            }
        }
    }
    __syncthreads();   // lse / delta / tab complete

    // ---------------- phase B: query-major scores, dK and dV of this wave's key tiles ----------------
    {
        bf16x8 kfB[QT], vfB[QT];
        int ktok[QT];
#pragma unroll
        for (int jk = 0; jk < QT; ++jk) {
            const int kt = wave * QT + jk;
            ktok[jk] = kt < NTILES ? tokoff[kt * 16 + fr] : -2;
            kfB[jk] = *(const bf16x8*)(rowp(ktok[jk]) + C + hoff + 8 * fg);
            vfB[jk] = *(const bf16x8*)(rowp(ktok[jk]) + 2 * C + hoff + 8 * fg);
        }
        f32x4 dk[2][QT], dv[2][QT];
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int jk = 0; jk < QT; ++jk) { dk[dt][jk] = zero4; dv[dt][jk] = zero4; }
#pragma unroll 1
        for (int qb = 0; qb < NKB; ++qb) {
            f32x4 pt[2][QT], dst[2][QT];
#pragma unroll
            for (int qi2 = 0; qi2 < 2; ++qi2) {
                const int qt = 2 * qb + qi2;
                if (qt < NTILES) {
                    const int qtk = tokoff[qt * 16 + fr];
                    const bf16x8 qa = *(const bf16x8*)(rowp(qtk) + hoff + 8 * fg);
                    const bf16x8 da = load_do(qtk);
                    const float4 l4 = *(const float4*)(lse + qt * 16 + 4 * fg);
                    const float4 d4 = *(const float4*)(delta + qt * 16 + 4 * fg);
                    const unsigned rq = *(const unsigned*)(rid + qt * 16 + 4 * fg);
                    const float lv[4] = {l4.x, l4.y, l4.z, l4.w}, dv4[4] = {d4.x, d4.y, d4.z, d4.w};
#pragma unroll
                    for (int jk = 0; jk < QT; ++jk) {
                        const int kt = wave * QT + jk;
                        const int ki = (kt < NTILES ? kt : 0) * 16 + fr;
                        const f32x4 sv = mfma16(qa, kfB[jk], zero4);     // S[q = 4fg+r][key = fr]
                        const f32x4 dp = mfma16(da, vfB[jk], zero4);
                        const float4 bb = *(const float4*)(p.bias_k + ((long)head * NP + ki) * NP + qt * 16 + 4 * fg);
                        const float bv[4] = {bb.x, bb.y, bb.z, bb.w};
                        const int ridk = rid[ki];
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            float v = sv[r] * p.scale + bv[r];
                            if (p.shift > 0 && (int)((rq >> (8 * r)) & 0xffu) != ridk) v -= 100.0f;
                            const float pr = (kt < NTILES) ? __expf(v - lv[r]) : 0.f;
                            pt[qi2][jk][r] = pr;
                            dst[qi2][jk][r] = pr * (dp[r] - dv4[r]);
                        }
                    }
                } else {
#pragma unroll
                    for (int jk = 0; jk < QT; ++jk) { pt[qi2][jk] = zero4; dst[qi2][jk] = zero4; }
                }
            }
            // dV^T[d][key] += dO^T[d][q] P[q][key] ;  dK^T[d][key] += Q^T[d][q] dS[q][key]
#pragma unroll
            for (int dt = 0; dt < 2; ++dt) {
                const bf16* orow = dOt + (dt * 16 + fr) * KP + 32 * qb + 4 * fg;
                const bf16* qrow = Qt + (dt * 16 + fr) * KP + 32 * qb + 4 * fg;
                const bf16x4 olo = *(const bf16x4*)orow, ohi = *(const bf16x4*)(orow + 16);
                const bf16x4 qlo = *(const bf16x4*)qrow, qhi = *(const bf16x4*)(qrow + 16);
                bf16x8 of, qf2;
#pragma unroll
                for (int r = 0; r < 4; ++r) { of[r] = olo[r]; of[4 + r] = ohi[r]; qf2[r] = qlo[r]; qf2[4 + r] = qhi[r]; }
#pragma unroll
                for (int jk = 0; jk < QT; ++jk) {
                    bf16x8 pb, db;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        pb[r] = (bf16)pt[0][jk][r]; pb[4 + r] = (bf16)pt[1][jk][r];
                        db[r] = (bf16)dst[0][jk][r]; db[4 + r] = (bf16)dst[1][jk][r];
                    }
                    dv[dt][jk] = mfma16(of, pb, dv[dt][jk]);
                    dk[dt][jk] = mfma16(qf2, db, dk[dt][jk]);
                }
            }
        }
#pragma unroll
        for (int jk = 0; jk < QT; ++jk) {
            const int kt = wave * QT + jk;
            if (kt >= NTILES || ktok[jk] == -2) continue;
#pragma unroll
            for (int dt = 0; dt < 2; ++dt) {
                const int col = hoff + dt * 16 + 4 * fg;
                if (ktok[jk] >= 0) {
                    bf16x4 kv, vv;
#pragma unroll
                    for (int r = 0; r < 4; ++r) { kv[r] = (bf16)(dk[dt][jk][r] * p.scale); vv[r] = (bf16)dv[dt][jk][r]; }
                    *(bf16x4*)(p.dqkv + (long)ktok[jk] * C3 + C + col) = kv;
                    *(bf16x4*)(p.dqkv + (long)ktok[jk] * C3 + 2 * C + col) = vv;
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        atomicAdd(p.dbias_pad + C + col + r, dk[dt][jk][r] * p.scale);
                        atomicAdd(p.dbias_pad + 2 * C + col + r, dv[dt][jk][r]);
                    }
                }
            }
        }
    }
    // relative-position table gradient of this (window, head): one contiguous burst
    for (int t = threadIdx.x; t < TT; t += NTH) atomicAdd(p.dtab + (long)head * TT + t, tab[t]);
}

// expanded relative-position bias: table ((2ws-1)^2, nH) fp32 -> bias_q [h][q][key], bias_k [h][key][q]
__global__ void relpos_expand_kernel(const float* __restrict__ table, float* __restrict__ bias_q, float* __restrict__ bias_k,
                                     int nH, int ws, int NP) {
    const int N = ws * ws, T1 = 2 * ws - 1;
    const long total = (long)nH * NP * NP;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int key = (int)(i % NP), q = (int)((i / NP) % NP), h = (int)(i / ((long)NP * NP));
        float v = 0.f;
        if (key >= N) v = -30000.0f;
        else if (q < N) {
            const int qy = q / ws, qx = q % ws, ky = key / ws, kx = key % ws;
            v = table[((qy - ky + ws - 1) * T1 + (qx - kx + ws - 1)) * nH + h];
        }
        bias_q[i] = v;
        bias_k[((long)h * NP + key) * NP + q] = v;
    }
}

static int wattn_ntiles(int ws) { return (ws * ws + 15) / 16; }

extern "C" int uenc_window_attn_np(int ws) { return wattn_ntiles(ws) * 16; }

extern "C" int uenc_relpos_expand(const float* table, float* bias_q, float* bias_k, int nH, int ws, hipStream_t stream) {
    UENC_CHECK_ARG(table && bias_q && bias_k && nH > 0 && ws > 0 && ws <= 12);
    const int NP = wattn_ntiles(ws) * 16;
    const long total = (long)nH * NP * NP;
    long blocks = (total + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(relpos_expand_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, table, bias_q, bias_k, nH, ws, NP);
    UENC_LAUNCH_RET();
}

static int fill_params(WAttn& p, const void* qkv, const void* qkv_bias, const float* bias_q, const float* bias_k, int B, int H,
                       int W, int C, int nH, int ws, int shift, float scale) {
    if (!(qkv && qkv_bias && bias_q && B > 0 && H > 0 && W > 0 && nH > 0 && C == nH * 32)) return UENC_EINVAL;
    if (!(ws >= 1 && ws <= 12 && shift >= 0 && shift < ws)) return UENC_EINVAL;
    if (((uintptr_t)qkv & 15) || ((uintptr_t)qkv_bias & 15)) return UENC_EINVAL;
    p.qkv = (const bf16*)qkv; p.qkv_bias = (const bf16*)qkv_bias; p.bias_q = bias_q; p.bias_k = bias_k;
    p.B = B; p.H = H; p.W = W; p.C = C; p.nH = nH; p.ws = ws; p.shift = shift;
    p.Hp = (H + ws - 1) / ws * ws; p.Wp = (W + ws - 1) / ws * ws;
    p.nWw = p.Wp / ws; p.nWin = (p.Hp / ws) * p.nWw; p.nWinTotal = B * p.nWin; p.N = ws * ws;
    p.scale = scale;
    p.out = nullptr; p.o_saved = nullptr; p.d_out = nullptr; p.dqkv = nullptr; p.dtab = nullptr; p.dbias_pad = nullptr;
    return UENC_OK;
}

template <int NT>
static void launch_fwd(const WAttn& p, hipStream_t stream) {
    const unsigned grid = (unsigned)((p.nWinTotal + 7) / 8 * 8 * p.nH);
    hipLaunchKernelGGL(wattn_fwd_kernel<NT>, dim3(grid), dim3(64 * WCfg<NT>::WAVES), 0, stream, p);
}
template <int NT>
static int launch_bwd(const WAttn& p, hipStream_t stream) {
    using Cf = WCfg<NT>;
    constexpr int NK2 = Cf::NKB * 32;
    const int T1 = 2 * p.ws - 1;
    const size_t shm = (size_t)((3 * 32 * Cf::KP * 2 + NK2 * 4 * 3 + NK2 * 2 + NK2 + 15) / 16 * 16) + (size_t)T1 * T1 * 4;
    const unsigned grid = (unsigned)((p.nWinTotal + 7) / 8 * 8 * p.nH);
    if (shm > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void*)wattn_bwd_kernel<NT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
        if (e != hipSuccess) return (int)e;
    }
    hipLaunchKernelGGL(wattn_bwd_kernel<NT>, dim3(grid), dim3(64 * Cf::WAVES), shm, stream, p);
    return UENC_OK;
}

#define WATTN_DISPATCH(NTV, CALL)                 \
    switch (NTV) {                                \
        case 1: CALL(1); break;                   \
        case 2: CALL(2); break;                   \
        case 3: CALL(3); break;                   \
        case 4: CALL(4); break;                   \
        case 5: CALL(5); break;                   \
        case 6: CALL(6); break;                   \
        case 7: CALL(7); break;                   \
        case 8: CALL(8); break;                   \
        case 9: CALL(9); break;                   \
        default: return UENC_EINVAL;              \
    }

extern "C" int uenc_window_attn_fwd(const void* qkv, const void* qkv_bias, const float* bias_q, void* out, int B, int H, int W,
                                    int C, int nH, int ws, int shift, float scale, hipStream_t stream) {
    WAttn p;
    int rc = fill_params(p, qkv, qkv_bias, bias_q, bias_q, B, H, W, C, nH, ws, shift, scale);
    if (rc != UENC_OK) return rc;
    UENC_CHECK_ARG(out != nullptr);
    p.out = (bf16*)out;
#define CALL(NT) launch_fwd<NT>(p, stream)
    WATTN_DISPATCH(wattn_ntiles(ws), CALL)
#undef CALL
    UENC_LAUNCH_RET();
}

extern "C" int uenc_window_attn_bwd(const void* qkv, const void* qkv_bias, const float* bias_q, const float* bias_k,
                                    const void* o_saved, const void* d_out, void* dqkv, float* dtab, float* dbias_pad, int B,
                                    int H, int W, int C, int nH, int ws, int shift, float scale, hipStream_t stream) {
    WAttn p;
    int rc = fill_params(p, qkv, qkv_bias, bias_q, bias_k, B, H, W, C, nH, ws, shift, scale);
    if (rc != UENC_OK) return rc;
    UENC_CHECK_ARG(bias_k && o_saved && d_out && dqkv && dtab && dbias_pad);
    p.o_saved = (const bf16*)o_saved; p.d_out = (const bf16*)d_out; p.dqkv = (bf16*)dqkv; p.dtab = dtab; p.dbias_pad = dbias_pad;
#define CALL(NT) { rc = launch_bwd<NT>(p, stream); if (rc != UENC_OK) return rc; }
    WATTN_DISPATCH(wattn_ntiles(ws), CALL)
#undef CALL
    UENC_LAUNCH_RET();
}
