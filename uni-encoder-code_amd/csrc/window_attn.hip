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
// One workgroup per (window, head), one wave per 16-token tile (9 waves for a 12x12 window).
//   1. every thread issues its share of the window's q/k/v(/dO) rows as 16-byte loads back to back
//      (the whole 27-36 KB of a window-head is in flight at once: HBM-latency is paid once), then
//      writes them row-major into LDS (64-byte rows, 16-byte chunks XOR-swizzled: conflict-free for
//      ds_read_b128 fragment reads and for ds_read_b64_tr_b16 transposed reads);
//   2. S^T[key][q] = K Q^T on MFMA 16x16x32 (head_dim 32 = one k-step): a lane owns one query
//      column, so softmax is register math + two shuffles; exp2 with log2(e) folded into scale/bias;
//   3. O^T[d][q] = V^T P^T: P^T tiles are the next MFMA's B operand without lane movement, V^T
//      fragments come from the row-major V image through the hardware transposing LDS read.
// Backward recomputes S in both orientations (key-major for dQ, query-major for dK / dV), so every
// gradient of a window is produced by its own workgroup; the relative-position-table gradient is
// accumulated in LDS and written as one partial row per workgroup (summed by the caller), the
// qkv-bias gradient contributed by padding slots uses atomics (few slots).
// Algorithmic HBM bytes per token per head: fwd read 3*64 + write 64; bwd read 5*64 + write 3*64.
#include "common.h"
#include "lds_frag.h"
#include <stdlib.h>
#include <type_traits>

#define LOG2E 1.4426950408889634f


struct WAttn {
    const bf16* qkv;        // (B, H, W, 3C)
    const bf16* qkv_bias;   // (3C) bf16 copy of attn.qkv.bias
    const float* bias_q;    // (nH, NP, NP) [h][q][key]  log2(e) * relative-position bias, -30000 for key >= N
    const float* bias_k;    // (nH, NP, NP) [h][key][q]  same, key-major (backward phase B)
    bf16* out;              // (B, H, W, C) attention output (before proj)
    float* lse;             // (B, H, W, nH) fp32 or NULL: log2-domain row statistics max + log2(sum) of the softmax -- written by the forward,
                            // read by the 12 x 12 backward (which then needs no max / sum passes); NULL: the backward recomputes them
    // backward only
    const bf16* o_saved;    // forward output
    const bf16* d_out;      // (B, H, W, C)
    bf16* dqkv;             // (B, H, W, 3C)
    float* dtab_ws;         // (nH * G, NP * NP) dense dS partial per workgroup (overwritten)
    float* dpad;            // (3C) q|k|v bias gradient that reaches the heads through padding slots: ADDED (atomics) to the caller's buffer
    int B, H, W, C, nH, ws, shift, Hp, Wp, nWw, nWin, nWinTotal, N;
    float scale;
    int variant;            // UENC_WATTN_VARIANT (A/B switches, 0 in production)
};

template <int NTILES>
struct WCfg {
    static constexpr int WAVES = NTILES;
    static constexpr int NTH = 64 * NTILES;
    static constexpr int NP = NTILES * 16;
    static constexpr int NKB = (NTILES + 1) / 2;   // 32-key blocks
    static constexpr int NK2 = NKB * 32;           // rows staged (zero / bias filled beyond N)
};

__device__ __forceinline__ int region3(int v, int P, int ws, int shift) { return (v >= P - ws) + (v >= P - shift); }

// slot bookkeeping: tokoff = flat token index, -1 padding slot, -2 beyond the window's N tokens
// WS12: the window size is the compile-time 12 of the 9-tile kernels (the per-slot divisions become multiplies)
template <int NK2, bool WS12 = false>
__device__ __forceinline__ void window_slots(const WAttn& p, int b, int wi, int wj, int* tokoff, unsigned char* rid,
                                             unsigned short* yx, int nthreads, int tid = -1) {
    const int ws = WS12 ? 12 : p.ws;
    for (int t = tid < 0 ? (int)threadIdx.x : tid; t < NK2; t += nthreads) {
        int off = -2, r = 0, code = 0;
        if (t < (WS12 ? 144 : p.N)) {
            const int ty = t / ws, tx = t - ty * ws;
            const int hs = wi * ws + ty, wx = wj * ws + tx;
            int ho = hs + p.shift, wo = wx + p.shift;
            if (ho >= p.Hp) ho -= p.Hp;
            if (wo >= p.Wp) wo -= p.Wp;
            off = (ho < p.H && wo < p.W) ? (b * p.H + ho) * p.W + wo : -1;
            if (p.shift > 0) r = 3 * region3(hs, p.Hp, ws, p.shift) + region3(wx, p.Wp, ws, p.shift);
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

// Bulk staging: NIMG images of NK2 rows x 4 chunks; all loads of a thread are issued before its stores.
//   src row pointer: tok >= 0 -> base[img] + tok * stride[img]; tok == -1 -> pad[img] (or zeros); tok == -2 -> zeros
template <int NIMG, int NK2, int NTH>
__device__ __forceinline__ void stage_images(unsigned char* const (&img)[NIMG], const bf16* const (&base)[NIMG],
                                             const long (&stride)[NIMG], const bf16* const (&pad)[NIMG], const int* tokoff) {
    constexpr int PER = (NK2 * 4 + NTH - 1) / NTH;      // chunks per thread per image (1-2)
    u32x4 v[NIMG][PER];
    int tok[PER];
#pragma unroll
    for (int it = 0; it < PER; ++it) {
        const int idx = threadIdx.x + it * NTH;
        tok[it] = idx < NK2 * 4 ? tokoff[idx >> 2] : -2;
    }
#pragma unroll
    for (int k = 0; k < NIMG; ++k)
#pragma unroll
        for (int it = 0; it < PER; ++it) {
            const int ch = (threadIdx.x + it * NTH) & 3;
            const bf16* src = tok[it] >= 0 ? base[k] + (long)tok[it] * stride[k] : pad[k];
            v[k][it] = (u32x4){0u, 0u, 0u, 0u};
            if (tok[it] >= 0 || (tok[it] == -1 && pad[k] != nullptr)) v[k][it] = *(const u32x4*)(src + ch * 8);
        }
#pragma unroll
    for (int k = 0; k < NIMG; ++k)
#pragma unroll
        for (int it = 0; it < PER; ++it) {
            const int idx = threadIdx.x + it * NTH;
            if (idx < NK2 * 4) *(u32x4*)(img[k] + rm_off(idx >> 2, idx & 3)) = v[k][it];
        }
}

// LDSB (12 x 12 windows): the bias comes from a 529-entry LDS copy of the head's table (gathered from the dense bias while the q / k / v
// rows are in flight) instead of 83 KB of dense [q][key] rows per window-head -- three times the bytes of its q, k and v.  Same values.
template <int NTILES, bool LDSB>
__global__ __launch_bounds__(64 * NTILES) void wattn_fwd_kernel(WAttn p) {
    static_assert(!LDSB || NTILES == 9, "the LDS bias table is laid out for ws = 12");
    using Cf = WCfg<NTILES>;
    constexpr int NTH = Cf::NTH, NP = Cf::NP, NKB = Cf::NKB, NK2 = Cf::NK2;
    constexpr int OFF_BT = (3 * NK2 * 64 + NK2 * 4 + NK2 + 15) / 16 * 16;      // LDSB: float rev[532], rev[528 - t] = log2(e) * table[t][head]
    __shared__ __attribute__((aligned(16))) unsigned char smem[OFF_BT + (LDSB ? 532 * 4 : 0)];
    unsigned char* Qs = smem;
    unsigned char* Ks = smem + NK2 * 64;
    unsigned char* Vs = smem + 2 * NK2 * 64;
    int* tokoff = (int*)(smem + 3 * NK2 * 64);
    unsigned char* rid = smem + 3 * NK2 * 64 + NK2 * 4;

    int win, head;
    decode_block(p, win, head);
    if (win >= p.nWinTotal) return;
    const int b = win / p.nWin, wrem = win - b * p.nWin;
    const int wi = wrem / p.nWw, wj = wrem - wi * p.nWw;
    const int lane = threadIdx.x & 63, qt = threadIdx.x >> 6, fr = lane & 15, fg = lane >> 4;
    const int C = p.C, hoff = head * 32;
    const long C3 = 3 * (long)p.C;

    float tabv = 0.f;                                  // LDSB: table entry t = threadIdx.x = (dy + 11) * 23 + dx + 11, from q = (max(dy,0), max(dx,0)), key = q - (dy,dx)
    if constexpr (LDSB) {
        if (threadIdx.x < 529) {
            const int t = threadIdx.x, dy = t / 23 - 11, dx = t % 23 - 11;
            const int qy = dy > 0 ? dy : 0, qx = dx > 0 ? dx : 0;
            tabv = p.bias_q[((long)head * NP + qy * 12 + qx) * NP + (qy - dy) * 12 + (qx - dx)];
        }
    }
    window_slots<NK2, NTILES == 9>(p, b, wi, wj, tokoff, rid, nullptr, NTH);
    __syncthreads();
    {
        unsigned char* const img[3] = {Qs, Ks, Vs};
        const bf16* const base[3] = {p.qkv + hoff, p.qkv + C + hoff, p.qkv + 2 * C + hoff};
        const long stride[3] = {C3, C3, C3};
        const bf16* const pad[3] = {p.qkv_bias + hoff, p.qkv_bias + C + hoff, p.qkv_bias + 2 * C + hoff};
        stage_images<3, NK2, NTH>(img, base, stride, pad, tokoff);
    }
    if constexpr (LDSB) {
        if (threadIdx.x < 529) ((float*)(smem + OFF_BT))[528 - threadIdx.x] = tabv;
    }
    __syncthreads();

    const bool masked = p.shift > 0 && (wi == p.Hp / p.ws - 1 || wj == p.nWw - 1);
    const int qi = qt * 16 + fr;
    const int qtok = tokoff[qi];
    const bf16x8 qf = frag_rows(Qs, qt * 16, fr, fg);
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    f32x4 s[NTILES];
#pragma unroll
    for (int kt = 0; kt < NTILES; ++kt) s[kt] = mfma16(frag_rows(Ks, kt * 16, fr, fg), qf, zero4);

    const float sc = p.scale * LOG2E;
    const float* brow = p.bias_q + ((long)head * NP + qi) * NP;
    const float* revq = (const float*)(smem + OFF_BT) + (528 - ((qi / 12 + 11) * 23 + qi % 12 + 11));     // + 23 ky + kx: the four keys of a lane are consecutive
    const int ridq = rid[qi];
    float mx = -1e30f;
#pragma unroll
    for (int kt = 0; kt < NTILES; ++kt) {
        float4 bb;
        if constexpr (LDSB) {
            const int key0 = kt * 16 + 4 * fg;
            const float* rb = revq + (key0 / 12) * 23 + key0 % 12;
            bb = make_float4(rb[0], rb[1], rb[2], rb[3]);
        } else {
            bb = *(const float4*)(brow + kt * 16 + 4 * fg);
        }
        s[kt][0] = s[kt][0] * sc + bb.x; s[kt][1] = s[kt][1] * sc + bb.y;
        s[kt][2] = s[kt][2] * sc + bb.z; s[kt][3] = s[kt][3] * sc + bb.w;
        if (masked) {
            const unsigned rk = *(const unsigned*)(rid + kt * 16 + 4 * fg);
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if ((int)((rk >> (8 * r)) & 0xffu) != ridq) s[kt][r] -= 100.0f * LOG2E;
        }
        mx = fmaxf(mx, fmaxf(fmaxf(s[kt][0], s[kt][1]), fmaxf(s[kt][2], s[kt][3])));
    }
    mx = xor16_max(mx);
    mx = xor32_max(mx);
    float sum = 0.f;
#pragma unroll
    for (int kt = 0; kt < NTILES; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float e = fast_exp2(s[kt][r] - mx);
            s[kt][r] = e;
            sum += e;
        }
    sum = xor16_sum(sum);
    sum = xor32_sum(sum);
    const float inv = 1.0f / sum;
    if (p.lse != nullptr && fg == 0 && qtok >= 0) p.lse[(long)qtok * p.nH + head] = mx + __builtin_amdgcn_logf(sum);     // v_log_f32 = log2

    f32x4 o[2] = {zero4, zero4};
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb) {
        bf16x8 pb;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            pb[r] = (bf16)s[2 * kb][r];
            pb[4 + r] = (2 * kb + 1 < NTILES) ? (bf16)s[(2 * kb + 1 < NTILES) ? 2 * kb + 1 : 0][r] : (bf16)0.f;
        }
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) o[dt] = mfma16(frag_tr(Vs, 32 * kb, 32 * kb + 16, dt * 16, lane), pb, o[dt]);
    }
    if (qtok >= 0) {
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
            bf16x4 ov;
#pragma unroll
            for (int r = 0; r < 4; ++r) ov[r] = (bf16)(o[dt][r] * inv);
            *(bf16x4*)(p.out + (long)qtok * C + hoff + dt * 16 + 4 * fg) = ov;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// backward
// ------------------------------------------------------------------------------------------------
// Persistent: the grid is nH x G workgroups, workgroup (head, g) walks windows g, g + G, ... of its head.
//   * the q/k/v/dO rows of the NEXT window go global -> LDS by LDS-DMA (global_load_lds_dwordx4, lane-linear destination,
//     XOR swizzle applied to the per-lane source chunk) into the second of two stages while the current window is
//     computed; its saved forward output rows (for delta) are prefetched into registers;
//   * the relative-position-table gradient is NOT scattered per window (LDS float atomics cost ~3 cycles per lane:
//     20k of them per window-head were two thirds of this kernel): dS is the same accumulator tile for every window, so
//     it is summed over the workgroup's windows in registers and written once as a dense [q][key] partial;
//     wattn_dtable_kernel then sums the partials over g and along the diagonals (q - key = const) into the table gradient.
__device__ __attribute__((aligned(16))) unsigned int g_wattn_zero16[4];      // a 16-byte chunk of zeros (rows beyond N, dO pad rows)
__device__ float g_wattn_big = 1e30f;                                        // "row statistic" of a slot without a token: exp2(s - 1e30) = 0

//   * LDSB (12 x 12 windows): the relative-position bias of phase A is looked up in a 529-entry LDS copy of the head's table
//     instead of being loaded as dense [q][key] rows from L2 (83 KB per window-head, more than its q / k / v / dO): an ordinary
//     global load whose result is used while the next window's LDS-DMA is in flight makes the wave wait for vmcnt(0), i.e. for
//     that whole DMA, in the middle of phase A (loads retire in order).  Same values, bit-identical results.
//   * LOADER: a TENTH wave does nothing but the slot bookkeeping and the LDS-DMA of the next window.  In-kernel s_memtime stamps
//     (profiles/r03_wattn_bwd_experiments.txt) showed every compute wave spending 16 % of a window inside its share of the DMA issue
//     (4-5 scattered 1 KB global_load_lds each, all nine waves through the CU's one address path at once) and 7 % in the slot
//     arithmetic, and then waiting vmcnt(0) for its own DMA in the middle of phase A (loads retire in order behind the DMA).  With
//     the loader the compute waves issue no DMA at all: their only vector-memory operations are the saved-output prefetch and the
//     result stores.  The loader takes part in the three barriers of a window: stage landed (its vmcnt(0)) -> top barrier -> slots +
//     DMA of the NEXT window into the other stage (free since the previous end barrier) -> mid barrier (the slots become visible:
//     the compute waves prefetch their next saved-output rows) -> end barrier.
//   * LSE (with LOADER): the forward saved the row statistics; the loader brings the 144 values of a window by three 4-byte-per-lane
//     LDS-DMA instructions and phase A computes P = exp2(s - lse) directly: no running maximum, no sum, no reciprocal, no cross-lane
//     butterflies (about a quarter of phase A's VALU instructions; the phases' time follows their VALU count).
template <int NTILES, bool LDSB, bool LOADER, bool LSE = false>
__global__ __launch_bounds__(64 * (NTILES + (LOADER ? 1 : 0))) void wattn_bwd_kernel(WAttn p, int G) {
    static_assert(!LDSB || NTILES == 9, "the LDS bias table is laid out for ws = 12 (N = NP = 144, key quads never straddle a window row)");
    static_assert(!LSE || LOADER, "the saved row statistics come in through the loader wave");
    using Cf = WCfg<NTILES>;
    constexpr int NWAVES = NTILES + (LOADER ? 1 : 0), NTH = 64 * NWAVES;
    constexpr int NP = Cf::NP, NKB = Cf::NKB, NK2 = Cf::NK2;
    constexpr int NIMG = LOADER ? 5 : 4;               // q, k, v, dO (+ LOADER: the saved forward output O, for delta)
    constexpr int IMG = NK2 * 64, STAGE = NIMG * IMG;
    constexpr int NTB = LOADER ? 3 : 2;                // slot buffers: LOADER keeps the window after next ready as well
    constexpr int OFF_TOK = 2 * STAGE;                 // int tokoff[NTB][NK2]
    constexpr int OFF_RID = OFF_TOK + NTB * NK2 * 4;   // u8  rid[NTB][NK2]
    constexpr int OFF_DEL = (OFF_RID + NTB * NK2 + 15) / 16 * 16;   // float delta[NK2]
    constexpr int OFF_PAD = OFF_DEL + NK2 * 4;         // float padacc[96]
    constexpr int OFF_P = OFF_PAD + 96 * 4;            // bf16 P[key tile][NK2 query rows][16 keys]: the softmax of phase A, read back
    constexpr int PSUB = NK2 * 32;                     //   transposed (ds_read_b64_tr_b16) as the P / dS operand tiles of phase B
    constexpr int OFF_BT = OFF_P + NTILES * PSUB;      // LDSB: float rev[532], rev[528 - t] = log2(e) * table[t][head]
    constexpr int OFF_LSE = OFF_BT + 532 * 4;          // LSE: float lse_s[2 stages][192]: the saved row statistics of the window's queries
    constexpr int OFF_MF = OFF_LSE + 1536;             // LOADER: int mflag[4]: "this window takes the shift mask", per slot buffer (written with the slots)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float* delta = (float*)(smem + OFF_DEL);
    float* padacc = (float*)(smem + OFF_PAD);          // [3][32] q|k|v bias gradient from padding slots, summed over this WG's windows
    unsigned char* Pimg = smem + OFF_P;

    // block -> (head, g): heads 2i / 2i+1 (the two halves of a 128-byte line of q, k, v, dO) on the same XCD
    int head, g;
    {
        const int x = blockIdx.x & 7, idx = blockIdx.x >> 3;
        const int U = (idx >> 1) * 8 + x, npair = (p.nH + 1) >> 1;
        head = 2 * (U % npair) + (idx & 1);
        g = U / npair;
    }
    if (head >= p.nH || g >= G) return;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), fr = lane & 15, fg = lane >> 4;
    const int C = p.C, hoff = head * 32, C3i = 3 * p.C;
    const long C3 = 3 * (long)p.C;
    const float sc = p.scale * LOG2E;
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    typedef __attribute__((address_space(3))) void lds_void;

    const bool loader = LOADER && wave == NTILES;
    auto slots = [&](int win, int st) {                // st: slot buffer
        const int b = win / p.nWin, wrem = win - b * p.nWin;
        const int wi = wrem / p.nWw, wj = wrem - wi * p.nWw;
        if (LOADER) {
            window_slots<NK2, NTILES == 9>(p, b, wi, wj, (int*)(smem + OFF_TOK) + st * NK2, smem + OFF_RID + st * NK2, nullptr, 64, lane);
            // the compute waves read the flag instead of repeating the three integer divisions of the window decode (~500 cycles per window)
            if (lane == 0) ((int*)(smem + OFF_MF))[st] = p.shift > 0 && (wi == p.Hp / p.ws - 1 || wj == p.nWw - 1);
        } else window_slots<NK2, NTILES == 9>(p, b, wi, wj, (int*)(smem + OFF_TOK) + st * NK2, smem + OFF_RID + st * NK2, nullptr, NTH);
    };
    // LDS-DMA of one window's q, k, v, dO images: 16 rows (1 KB) per wave instruction
    auto issue = [&](int st, int tb) {                 // st: stage, tb: slot buffer
        const int* tok = (const int*)(smem + OFF_TOK) + tb * NK2;
        constexpr int PER_IMG = NK2 / 16;
        if constexpr (LOADER) {
            // The loader wave: all five images of a 16-row block from one slot read.  Sources are a UNIFORM base per image plus a 32-bit
            // byte offset per lane (scalar-base addressing: no 64-bit address arithmetic per instruction), token rows and padding slots
            // as two exec-masked instruction groups (the second one is skipped by all but the windows on the map's edges); the rows
            // beyond the window's 144 slots were zeroed once at kernel start and are never written.  ~6 instructions per DMA instead
            // of ~25: the issue of a window's 45 DMAs took 8.3 k cycles before, longer than phase A.
            const char* bq = (const char*)(p.qkv + hoff);
            const char* bdo = (const char*)(p.d_out + hoff);
            const char* bo = (const char*)(p.o_saved + hoff);
            const char* bpad = (const char*)(p.qkv_bias + hoff);
            unsigned char* stage = smem + st * STAGE;
#pragma unroll 3
            for (int rb = 0; rb < NP / 16; ++rb) {
                const int row = rb * 16 + (lane >> 2), chunk = (lane & 3) ^ ((row >> 1) & 3);
                const int t = tok[row];
                unsigned char* d0 = stage + rb * 1024;
                if (t >= 0) {
                    const unsigned o3 = ((unsigned)(t * C3i) + chunk * 8) * 2u, o1 = ((unsigned)(t * C) + chunk * 8) * 2u;
                    __builtin_amdgcn_global_load_lds(bq + o3, (lds_void*)(d0), 16, 0, 0);
                    __builtin_amdgcn_global_load_lds(bq + 2 * C + o3, (lds_void*)(d0 + IMG), 16, 0, 0);
                    __builtin_amdgcn_global_load_lds(bq + 4 * C + o3, (lds_void*)(d0 + 2 * IMG), 16, 0, 0);
                    __builtin_amdgcn_global_load_lds(bdo + o1, (lds_void*)(d0 + 3 * IMG), 16, 0, 0);
                    __builtin_amdgcn_global_load_lds(bo + o1, (lds_void*)(d0 + 4 * IMG), 16, 0, 0);
                } else {                               // padding slot (t == -1; slots beyond N do not occur for 12 x 12): q, k, v = the qkv bias, dO = O = 0
                    const unsigned oc = chunk * 16u;
                    __builtin_amdgcn_global_load_lds(bpad + oc, (lds_void*)(d0), 16, 0, 0);
                    __builtin_amdgcn_global_load_lds(bpad + 2 * C + oc, (lds_void*)(d0 + IMG), 16, 0, 0);
                    __builtin_amdgcn_global_load_lds(bpad + 4 * C + oc, (lds_void*)(d0 + 2 * IMG), 16, 0, 0);
                    __builtin_amdgcn_global_load_lds((const char*)g_wattn_zero16, (lds_void*)(d0 + 3 * IMG), 16, 0, 0);
                    __builtin_amdgcn_global_load_lds((const char*)g_wattn_zero16, (lds_void*)(d0 + 4 * IMG), 16, 0, 0);
                }
            }
            if constexpr (LSE) {                       // one float per query slot: 64 slots (256 B) per instruction; padding slots get +inf-like (P = 0)
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    const int row = j * 64 + lane;
                    const int t = row < NK2 ? tok[row] : -2;
                    const float* src = t >= 0 ? p.lse + (long)t * p.nH + head : &g_wattn_big;
                    __builtin_amdgcn_global_load_lds(src, (lds_void*)(smem + OFF_LSE + (st * 192 + j * 64) * 4), 4, 0, 0);
                }
            }
        } else {
            for (int ii = wave; ii < 4 * PER_IMG; ii += NTILES) {
                const int img = ii / PER_IMG, rb = ii - img * PER_IMG;
                const int row = rb * 16 + (lane >> 2), chunk = (lane & 3) ^ ((row >> 1) & 3);
                const int t = tok[row];
                const bf16* src = (const bf16*)g_wattn_zero16;
                if (t >= 0) src = (img < 3 ? p.qkv + (long)t * C3 + img * C : p.d_out + (long)t * C) + hoff + chunk * 8;
                else if (t == -1 && img < 3) src = p.qkv_bias + img * C + hoff + chunk * 8;
                __builtin_amdgcn_global_load_lds(src, (lds_void*)(smem + st * STAGE + img * IMG + rb * 1024), 16, 0, 0);
            }
        }
    };

    // LDSB: entry t = (dy + 11) * 23 + dx + 11 of the head's table, gathered from the dense bias (q = (max(dy,0), max(dx,0)), key = q - (dy,dx))
    // and stored REVERSED: the four keys 4 fg .. 4 fg + 3 of a lane sit in one window row, so their entries are four consecutive floats
    const float* rev = (const float*)(smem + OFF_BT);
    int jb[NTILES];                                    // rev index of (this lane's query, key 16 kt + 4 fg): the same for every window
    if constexpr (LDSB) {
        for (int t = threadIdx.x; t < 529; t += NTH) {
            const int dy = t / 23 - 11, dx = t % 23 - 11;
            const int qy = dy > 0 ? dy : 0, qx = dx > 0 ? dx : 0;
            ((float*)(smem + OFF_BT))[528 - t] = p.bias_q[((long)head * NP + qy * 12 + qx) * NP + (qy - dy) * 12 + (qx - dx)];
        }
        const int qi0 = wave * 16 + fr, aq = 528 - ((qi0 / 12 + 11) * 23 + qi0 % 12 + 11);
#pragma unroll
        for (int kt = 0; kt < NTILES; ++kt) {
            const int key0 = kt * 16 + 4 * fg;
            jb[kt] = aq + (key0 / 12) * 23 + key0 % 12;
        }
    }
    f32x4 dsacc[NTILES];
#pragma unroll
    for (int kt = 0; kt < NTILES; ++kt) dsacc[kt] = zero4;
    for (int t = threadIdx.x; t < 96; t += NTH) padacc[t] = 0.f;
    for (int t = threadIdx.x; t < NTILES * PSUB / 16; t += NTH) ((u32x4*)Pimg)[t] = (u32x4){0u, 0u, 0u, 0u};     // rows >= NP are never written
    if constexpr (LOADER) {                            // rows NP .. NK2 - 1 of every staged image: zero for good (the loader writes rows < NP only)
        constexpr int TAIL = (NK2 - NP) * 4;           // 16-byte chunks per image
        for (int t = threadIdx.x; t < 2 * NIMG * TAIL; t += NTH)
            *(u32x4*)(smem + (t / TAIL) * IMG + NP * 64 + (t % TAIL) * 16) = (u32x4){0u, 0u, 0u, 0u};
    }

    int win = g;
    bf16x8 o_next;                                     // saved forward output row of this lane's query, next window
    {
#pragma unroll
        for (int j = 0; j < 8; ++j) o_next[j] = (bf16)0.f;
    }
    if (win < p.nWinTotal) {
        if (LOADER) {
            if (loader) {
                slots(win, 0);                         // written and read back by the same wave: LDS operations of a wave execute in order
                issue(0, 0);
                if (win + G < p.nWinTotal) slots(win + G, 1);
            }
            __syncthreads();
        } else {
            slots(win, 0);
            __syncthreads();
            issue(0, 0);
        }
        if (!LOADER) {
            const int qtok = ((const int*)(smem + OFF_TOK))[wave * 16 + fr];
            if (qtok >= 0) o_next = *(const bf16x8*)(p.o_saved + (long)qtok * C + hoff + 8 * fg);
        }
    }
    const bool wait_at_top = (p.variant & 1) != 0;          // bit 0: the round-2 placement of the stage wait (A/B)
    // LOADER: the waves talk through LDS only, so a barrier needs this wave's LDS operations done (lgkmcnt) and nothing else; __syncthreads()
    // would also drain vmcnt wherever an LDS-DMA may be outstanding -- the loader would wait for the DMA it has just issued and the compute
    // waves for the acknowledgement of their result stores, at every barrier
    auto wg_barrier = [&]() {
        if (LOADER) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        else __syncthreads();
    };
    // The loader's loop is separate code: with the LDS-DMA in the compute loop's control flow the compiler must assume one is outstanding
    // at every LDS read it cannot prove disjoint (every transposing read) and makes the compute waves wait vmcnt(0) there -- for
    // their own freshly issued result stores.
    if (loader) {
        for (int it = 0; win < p.nWinTotal; ++it, win += G) {
            // slot buffers rotate over three windows: the DMA of window it + 1 (~50 instructions, ~7.5 k cycles: as long as phase A) starts
            // right behind the top barrier from slots made ready a window earlier; the slots of window it + 2 are computed afterwards
            const int tb = it % 3, tb1 = tb == 2 ? 0 : tb + 1, tb2 = tb1 == 2 ? 0 : tb1 + 1;
            __builtin_amdgcn_s_waitcnt(0x0f70);            // the current window's images have landed
            wg_barrier();
            if (win + G < p.nWinTotal) issue((it & 1) ^ 1, tb1);
            wg_barrier();
            if (win + 2 * G < p.nWinTotal) slots(win + 2 * G, tb2);     // buffer of window it - 1: free since the previous end barrier
            wg_barrier();
        }
    } else
    for (int it = 0; win < p.nWinTotal; ++it, win += G) {
        const int st = it & 1;
        const unsigned char* Qs = smem + st * STAGE;
        const unsigned char* Ks = Qs + IMG;
        const unsigned char* Vs = Qs + 2 * IMG;
        const unsigned char* dOs = Qs + 3 * IMG;
        const int tbuf = LOADER ? it % 3 : st;
        const int* tokoff = (const int*)(smem + OFF_TOK) + tbuf * NK2;
        const unsigned char* rid = smem + OFF_RID + tbuf * NK2;
        bool masked;
        if constexpr (LOADER) {
            masked = __builtin_amdgcn_readfirstlane(((const int*)(smem + OFF_MF))[it % 3]) != 0;
        } else {
            const int b = win / p.nWin, wrem = win - b * p.nWin;
            const int wi = wrem / p.nWw, wj = wrem - wi * p.nWw;
            masked = p.shift > 0 && (wi == p.Hp / p.ws - 1 || wj == p.nWw - 1);
        }
        const bool more = win + G < p.nWinTotal;

        // bias rows come from L2 in groups of 3 key tiles, one group ahead of their use
        // (all NTILES at once would be 36 more live registers than the 168 a 9-wave workgroup can have)
        const float* brow0 = p.bias_q + ((long)head * NP + wave * 16 + fr) * NP + 4 * fg;
        float4 bnext[3];
        if constexpr (!LDSB) {
#pragma unroll
            for (int j = 0; j < 3; ++j) bnext[j] = *(const float4*)(brow0 + (j < NTILES ? j : 0) * 16);
        }
        if (!LOADER && more) slots(win + G, st ^ 1);
        // This wave's share of the current stage has landed: for the first window by the wait here; for the others by the wait in
        // front of the previous window's dK / dV stores (below).  A wave's vector-memory operations retire in order, so a vmcnt(0)
        // HERE would also wait for those stores to be acknowledged -- with one workgroup per CU nothing else runs meanwhile.
        if (!LOADER && (it == 0 || wait_at_top)) __builtin_amdgcn_s_waitcnt(0x0f70);
        const bf16x8 ov_reg = o_next;
        wg_barrier();                                  // ... everyone's; next window's slots are visible
        if (!LOADER && more) {
            issue(st ^ 1, st ^ 1);
            const int qtok_n = ((const int*)(smem + OFF_TOK))[(st ^ 1) * NK2 + wave * 16 + fr];
            if (qtok_n >= 0) o_next = *(const bf16x8*)(p.o_saved + (long)qtok_n * C + hoff + 8 * fg);
        }

        // ---------------- phase A: this wave's 16 queries x all keys (key-major S^T): statistics, dS, dQ ----------------
        {
            const int qt = wave, qi = qt * 16 + fr;
            const int qtok = tokoff[qi];
            const bf16x8 qf = frag_rows(Qs, qt * 16, fr, fg);
            const bf16x8 dof = frag_rows(dOs, qt * 16, fr, fg);
            const bf16x8 ov = LOADER ? frag_rows(Qs + 4 * IMG, qt * 16, fr, fg) : ov_reg;      // the same 8 channels of the saved output row (zeros for padding slots)
            float dl = 0.f;                               // delta[q] = sum_d dO[q][d] * O[q][d]
            if (qtok >= 0) {
#pragma unroll
                for (int j = 0; j < 8; ++j) dl += (float)ov[j] * (float)dof[j];
            }
            dl = xor16_sum(dl);
            dl = xor32_sum(dl);
            f32x4 s[NTILES];
            const int ridq = rid[qi];
            float mx = LSE ? ((const float*)(smem + OFF_LSE))[st * 192 + qi] : -1e30f;      // LSE: the saved max + log2(sum) takes the maximum's place
#pragma unroll
            for (int k3 = 0; k3 < NTILES; k3 += 3) {
                float4 bcur[3];
                if constexpr (LDSB) {
#pragma unroll
                    for (int j = 0; j < 3; ++j) {
                        const float* rb = rev + jb[k3 + j < NTILES ? k3 + j : 0];
                        bcur[j] = make_float4(rb[0], rb[1], rb[2], rb[3]);
                    }
                } else {
#pragma unroll
                    for (int j = 0; j < 3; ++j) bcur[j] = bnext[j];
                    if (k3 + 3 < NTILES) {
#pragma unroll
                        for (int j = 0; j < 3; ++j) bnext[j] = *(const float4*)(brow0 + (k3 + 3 + j < NTILES ? k3 + 3 + j : 0) * 16);
                    }
                }
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    const int kt = k3 + j;
                    if (kt < NTILES) {
                        s[kt] = mfma16(frag_rows(Ks, kt * 16, fr, fg), qf, zero4);
                        const float4 bb = bcur[j];
                        s[kt][0] = s[kt][0] * sc + bb.x; s[kt][1] = s[kt][1] * sc + bb.y;
                        s[kt][2] = s[kt][2] * sc + bb.z; s[kt][3] = s[kt][3] * sc + bb.w;
                        if (masked) {
                            const unsigned rk = *(const unsigned*)(rid + kt * 16 + 4 * fg);
#pragma unroll
                            for (int r = 0; r < 4; ++r)
                                if ((int)((rk >> (8 * r)) & 0xffu) != ridq) s[kt][r] -= 100.0f * LOG2E;
                        }
                        if (!LSE) mx = fmaxf(mx, fmaxf(fmaxf(s[kt][0], s[kt][1]), fmaxf(s[kt][2], s[kt][3])));
                    }
                }
                __builtin_amdgcn_sched_barrier(0);     // bound the loads hoisted ahead (register pressure)
            }
            if (!LSE) {
                mx = xor16_max(mx);
                mx = xor32_max(mx);
            }
            float sum = 0.f;
#pragma unroll
            for (int kt = 0; kt < NTILES; ++kt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float e = fast_exp2(s[kt][r] - mx);
                    s[kt][r] = e;
                    sum += e;
                }
            float inv = 1.0f;                              // LSE: the exponentials above are the probabilities already
            if (!LSE) {
                sum = xor16_sum(sum);
                sum = xor32_sum(sum);
                inv = 1.0f / sum;
            }
            if (fg == 0) delta[qi] = dl;
            __builtin_amdgcn_sched_barrier(0);
            const float qreal = qi < p.N ? 1.f : 0.f;
#pragma unroll
            for (int kt = 0; kt < NTILES; ++kt) {
                const f32x4 dp = mfma16(frag_rows(Vs, kt * 16, fr, fg), dof, zero4);     // dP^T[key][q] = V[key] . dO[q]
                bf16x4 p4;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float pn = s[kt][r] * inv;         // P[q][key]
                    p4[r] = (bf16)pn;
                    const float ds = pn * (dp[r] - dl);
                    s[kt][r] = ds;
                    dsacc[kt][r] += ds * qreal;              // (keys >= N have P = 0 exactly; queries >= N are zeroed here)
                }
                // key tile kt, row q, keys 4 fg .. 4 fg + 3.  The 8-byte chunk position inside the 32-byte row is XOR-ed with (q >> 2) & 3: the
                // sixteen lanes of a write (q = 16 w + 0..15, one fg) then cover all 32 banks -- un-swizzled they hit four 32-byte-strided
                // positions four deep (all of this kernel's SQ_LDS_BANK_CONFLICT cycles, ~8 % of its LDS time)
                *(bf16x4*)(Pimg + kt * PSUB + qi * 32 + ((fg ^ ((qi >> 2) & 3)) << 3)) = p4;
                if (kt % 3 == 2) __builtin_amdgcn_sched_barrier(0);
            }
            f32x4 dq[2] = {zero4, zero4};                  // dQ^T[d][q] = scale * sum_key K^T[d][key] dS^T[key][q]
#pragma unroll
            for (int kb = 0; kb < NKB; ++kb) {
                bf16x8 pb;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    pb[r] = (bf16)s[2 * kb][r];
                    pb[4 + r] = (2 * kb + 1 < NTILES) ? (bf16)s[(2 * kb + 1 < NTILES) ? 2 * kb + 1 : 0][r] : (bf16)0.f;
                }
#pragma unroll
                for (int dt = 0; dt < 2; ++dt) dq[dt] = mfma16(frag_tr(Ks, 32 * kb, 32 * kb + 16, dt * 16, lane), pb, dq[dt]);
            }
            if (qtok != -2) {
#pragma unroll
                for (int dt = 0; dt < 2; ++dt) {
                    const int col = hoff + dt * 16 + 4 * fg;
                    if (qtok >= 0) {
                        bf16x4 o4;
#pragma unroll
                        for (int r = 0; r < 4; ++r) o4[r] = (bf16)(dq[dt][r] * p.scale);
                        *(bf16x4*)(p.dqkv + (long)qtok * C3 + col) = o4;
                    } else {
#pragma unroll
                        for (int r = 0; r < 4; ++r) atomicAdd(padacc + dt * 16 + 4 * fg + r, dq[dt][r] * p.scale);
                    }
                }
            }
        }
        wg_barrier();      // delta and the P image complete

        // ---------------- phase B: this wave's 16 keys x all queries: dK, dV ----------------
        // P[q][key] comes back from LDS already in MFMA operand order (hardware transposing read), which is also the accumulator
        // order of dP = dO V^T: dS = P * (dP - delta) needs no score recomputation (no QK^T, bias, mask or exp here)
        {
            const int kt = wave, ki = kt * 16 + fr;
            const int ktok = tokoff[ki];
            const bf16x8 vfB = frag_rows(Vs, kt * 16, fr, fg);
            const unsigned char* Psub = Pimg + kt * PSUB;
            f32x4 dk[2] = {zero4, zero4}, dv[2] = {zero4, zero4};
#pragma unroll
            for (int qb = 0; qb < NKB; ++qb) {
                typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4;
                const int prow = 32 * qb + 4 * fg + (fr >> 2), pcol = ((fr & 3) ^ fg) * 8;      // (the row's chunk swizzle: (prow >> 2) & 3 == fg, also for prow + 16)
                const bf16x4 plo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(Psub + prow * 32 + pcol));
                const bf16x4 phi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(Psub + (prow + 16) * 32 + pcol));
                bf16x8 pb, db;
#pragma unroll
                for (int r = 0; r < 4; ++r) { pb[r] = plo[r]; pb[4 + r] = phi[r]; }
#pragma unroll
                for (int h2 = 0; h2 < 2; ++h2) {
                    const int qt = 2 * qb + h2;
                    if (qt < NTILES) {
                        const f32x4 dp = mfma16(frag_rows(dOs, qt * 16, fr, fg), vfB, zero4);    // dP[q = 4fg+r][key = fr]
                        const float4 d4 = *(const float4*)(delta + qt * 16 + 4 * fg);
                        db[4 * h2 + 0] = (bf16)((float)pb[4 * h2 + 0] * (dp[0] - d4.x));
                        db[4 * h2 + 1] = (bf16)((float)pb[4 * h2 + 1] * (dp[1] - d4.y));
                        db[4 * h2 + 2] = (bf16)((float)pb[4 * h2 + 2] * (dp[2] - d4.z));
                        db[4 * h2 + 3] = (bf16)((float)pb[4 * h2 + 3] * (dp[3] - d4.w));
                    } else {
#pragma unroll
                        for (int r = 0; r < 4; ++r) db[4 * h2 + r] = (bf16)0.f;
                    }
                }
#pragma unroll
                for (int dt = 0; dt < 2; ++dt) {
                    dv[dt] = mfma16(frag_tr(dOs, 32 * qb, 32 * qb + 16, dt * 16, lane), pb, dv[dt]);   // dV^T += dO^T P
                    dk[dt] = mfma16(frag_tr(Qs, 32 * qb, 32 * qb + 16, dt * 16, lane), db, dk[dt]);    // dK^T += Q^T dS
                }
            }
            // the next window's q / k / v / dO rows (LDS-DMA issued at the top of this iteration) have had both phases to land: waiting for
            // them here, BEFORE this window's last stores enter the queue, costs nothing and leaves the stores to drain under the next
            // window's phase A
            if (!LOADER && !wait_at_top) __builtin_amdgcn_s_waitcnt(0x0f70);
            if (ktok != -2) {
#pragma unroll
                for (int dt = 0; dt < 2; ++dt) {
                    const int col = hoff + dt * 16 + 4 * fg;
                    if (ktok >= 0) {
                        bf16x4 kv, vv;
#pragma unroll
                        for (int r = 0; r < 4; ++r) { kv[r] = (bf16)(dk[dt][r] * p.scale); vv[r] = (bf16)dv[dt][r]; }
                        *(bf16x4*)(p.dqkv + (long)ktok * C3 + C + col) = kv;
                        *(bf16x4*)(p.dqkv + (long)ktok * C3 + 2 * C + col) = vv;
                    } else {
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            atomicAdd(padacc + 32 + dt * 16 + 4 * fg + r, dk[dt][r] * p.scale);
                            atomicAdd(padacc + 64 + dt * 16 + 4 * fg + r, dv[dt][r]);
                        }
                    }
                }
            }
        }
        wg_barrier();      // stage st, lse / delta and tokoff[st] are free for the window after next
    }
    // dense dS partial of this workgroup: [wave = query tile][kt][lane][4]  <->  q = 16 wave + (lane & 15), key = 16 kt + 4 (lane >> 4) + r
    if (!loader) {
        float* wsp = p.dtab_ws + (((long)head * G + g) * NTILES + wave) * (NP * 16);
#pragma unroll
        for (int kt = 0; kt < NTILES; ++kt) *(f32x4*)(wsp + (kt * 64 + lane) * 4) = dsacc[kt];
    }
    for (int t = threadIdx.x; t < 96; t += NTH) {
        const float v = padacc[t];
        if (v != 0.f) atomicAdd(p.dpad + (t >> 5) * C + hoff + (t & 31), v);
    }
}

// Table gradient from the dense dS partials: one workgroup per (head, query tile) sums the partials of the G groups
// (coalesced) into LDS, then every table entry (dy, dx) adds up the slab's pairs q - key = (dy, dx).
template <int NTILES>
__global__ __launch_bounds__(256) void wattn_dtable_kernel(const float* __restrict__ wsd, float* __restrict__ dtab, int G, int nH,
                                                           int ws) {
    constexpr int NP = NTILES * 16, SLAB = NP * 16;
    __shared__ __attribute__((aligned(16))) float slab[SLAB];
    const int head = blockIdx.x / NTILES, qt = blockIdx.x - head * NTILES;
    for (int e = threadIdx.x; e < SLAB / 4; e += 256) {
        f32x4 a = {0.f, 0.f, 0.f, 0.f};
        for (int g = 0; g < G; ++g) a += *(const f32x4*)(wsd + (((long)head * G + g) * NTILES + qt) * SLAB + e * 4);
        *(f32x4*)(slab + e * 4) = a;
    }
    __syncthreads();
    const int N = ws * ws, T1 = 2 * ws - 1;
    for (int t = threadIdx.x; t < T1 * T1; t += 256) {
        const int dy = t / T1 - (ws - 1), dx = t % T1 - (ws - 1);
        float acc = 0.f;
        for (int fr = 0; fr < 16; ++fr) {
            const int q = qt * 16 + fr;
            if (q >= N) break;
            const int ky = q / ws - dy, kx = q % ws - dx;
            if (ky < 0 || ky >= ws || kx < 0 || kx >= ws) continue;
            const int key = ky * ws + kx;
            acc += slab[((key >> 4) * 64 + ((key & 15) >> 2) * 16 + fr) * 4 + (key & 3)];
        }
        if (acc != 0.f) atomicAdd(dtab + (long)t * nH + head, acc);        // the parameter's own layout ((2ws-1)^2, nH), accumulated
    }
}

// Grouped form: the table gradients of MANY window-attention backward passes in one launch (24 per step for Swin-L, ~19 us each alone).
// table: n descriptors in device memory, 40 bytes each: { const float* wsd; float* dtab; int G, nH, ws, ntiles, blk_begin, pad; }
// blk_begin = exclusive prefix sum of nH * ntiles; total_blocks = the full sum.  wsd = the dense dS partials uenc_window_attn_bwd wrote
// (defer_dtable = 1 leaves them un-reduced); they must stay untouched until this runs.
struct DtableDesc { const float* wsd; float* dtab; int G, nH, ws, ntiles, blk_begin, pad; };
__global__ __launch_bounds__(256) void wattn_dtable_grouped_kernel(const DtableDesc* __restrict__ table, int n) {
    __shared__ __attribute__((aligned(16))) float slab[9 * 16 * 16];       // NP * 16 for NTILES <= 9
    int lo = 0, hi = n - 1;
    const int blk = blockIdx.x;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (table[mid].blk_begin <= blk) lo = mid; else hi = mid - 1;
    }
    const DtableDesc d = table[lo];
    const int NT = d.ntiles, NP = NT * 16, SLAB = NP * 16;
    const int local = blk - d.blk_begin;
    const int head = local / NT, qt = local - head * NT;
    for (int e = threadIdx.x; e < SLAB / 4; e += 256) {
        f32x4 a = {0.f, 0.f, 0.f, 0.f};
        for (int g = 0; g < d.G; ++g) a += *(const f32x4*)(d.wsd + (((long)head * d.G + g) * NT + qt) * SLAB + e * 4);
        *(f32x4*)(slab + e * 4) = a;
    }
    __syncthreads();
    const int ws = d.ws, N = ws * ws, T1 = 2 * ws - 1;
    for (int t = threadIdx.x; t < T1 * T1; t += 256) {
        const int dy = t / T1 - (ws - 1), dx = t % T1 - (ws - 1);
        float acc = 0.f;
        for (int fr = 0; fr < 16; ++fr) {
            const int q = qt * 16 + fr;
            if (q >= N) break;
            const int ky = q / ws - dy, kx = q % ws - dx;
            if (ky < 0 || ky >= ws || kx < 0 || kx >= ws) continue;
            const int key = ky * ws + kx;
            acc += slab[((key >> 4) * 64 + ((key & 15) >> 2) * 16 + fr) * 4 + (key & 3)];
        }
        if (acc != 0.f) atomicAdd(d.dtab + (long)t * d.nH + head, acc);
    }
}

extern "C" int uenc_window_attn_dtable_grouped(const void* table, int n, int total_blocks, hipStream_t stream) {
    UENC_CHECK_ARG(table && n > 0 && total_blocks > 0 && ((uintptr_t)table & 7) == 0);
    static_assert(sizeof(DtableDesc) == 40, "descriptor layout is part of the ABI");
    hipLaunchKernelGGL(wattn_dtable_grouped_kernel, dim3((unsigned)total_blocks), dim3(256), 0, stream, (const DtableDesc*)table, n);
    UENC_LAUNCH_RET();
}

// expanded relative-position bias: table ((2ws-1)^2, nH) fp32 -> log2(e)-scaled bias_q [h][q][key], bias_k [h][key][q]
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
            v = LOG2E * table[((qy - ky + ws - 1) * T1 + (qx - kx + ws - 1)) * nH + h];
        }
        bias_q[i] = v;
        bias_k[((long)h * NP + key) * NP + q] = v;
    }
}

// Grouped form: the expanded biases of MANY attention modules in one launch (24 per step for Swin-L, ~9 us each alone for 1-4 MB of
// stores: the launches, not the bytes, were the cost).  table: n descriptors in device memory, 40 bytes each:
// { const float* table; float* bias_q; float* bias_k (may be NULL: not written); int nH, ws, NP, blk_begin; }, blk_begin = exclusive
// prefix sum of ceil(nH * NP * NP / 2048); total_blocks = the full sum.
struct RelposDesc { const float* table; float* bias_q; float* bias_k; int nH, ws, NP, blk_begin; };
__global__ __launch_bounds__(256) void relpos_expand_grouped_kernel(const RelposDesc* __restrict__ tab, int n) {
    int lo = 0, hi = n - 1;
    const int blk = blockIdx.x;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (tab[mid].blk_begin <= blk) lo = mid; else hi = mid - 1;
    }
    const RelposDesc d = tab[lo];
    const int ws = d.ws, NP = d.NP, N = ws * ws, T1 = 2 * ws - 1;
    const long total = (long)d.nH * NP * NP;
    const long base = (long)(blk - d.blk_begin) * 2048;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const long i = base + j * 256 + threadIdx.x;
        if (i >= total) break;
        const int key = (int)(i % NP), q = (int)((i / NP) % NP), h = (int)(i / ((long)NP * NP));
        float v = 0.f;
        if (key >= N) v = -30000.0f;
        else if (q < N) {
            const int qy = q / ws, qx = q % ws, ky = key / ws, kx = key % ws;
            v = LOG2E * d.table[((qy - ky + ws - 1) * T1 + (qx - kx + ws - 1)) * d.nH + h];
        }
        d.bias_q[i] = v;
        if (d.bias_k != nullptr) d.bias_k[((long)h * NP + key) * NP + q] = v;
    }
}

extern "C" int uenc_relpos_expand_grouped(const void* table, int n, int total_blocks, hipStream_t stream) {
    UENC_CHECK_ARG(table && n > 0 && total_blocks > 0 && ((uintptr_t)table & 7) == 0);
    static_assert(sizeof(RelposDesc) == 40, "descriptor layout is part of the ABI");
    hipLaunchKernelGGL(relpos_expand_grouped_kernel, dim3((unsigned)total_blocks), dim3(256), 0, stream, (const RelposDesc*)table, n);
    UENC_LAUNCH_RET();
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
    if ((long)B * H * W * 3 * C >= (1L << 31)) return UENC_EINVAL;     // 32-bit token offsets
    p.qkv = (const bf16*)qkv; p.qkv_bias = (const bf16*)qkv_bias; p.bias_q = bias_q; p.bias_k = bias_k;
    p.B = B; p.H = H; p.W = W; p.C = C; p.nH = nH; p.ws = ws; p.shift = shift;
    p.Hp = (H + ws - 1) / ws * ws; p.Wp = (W + ws - 1) / ws * ws;
    p.nWw = p.Wp / ws; p.nWin = (p.Hp / ws) * p.nWw; p.nWinTotal = B * p.nWin; p.N = ws * ws;
    p.scale = scale;
    { const char* e = getenv("UENC_WATTN_VARIANT"); p.variant = e ? atoi(e) : 0; }
    p.out = nullptr; p.lse = nullptr; p.o_saved = nullptr; p.d_out = nullptr; p.dqkv = nullptr; p.dtab_ws = nullptr; p.dpad = nullptr;
    return UENC_OK;
}

template <int NT>
static void launch_fwd(const WAttn& p, hipStream_t stream) {
    const unsigned grid = (unsigned)((p.nWinTotal + 7) / 8 * 8 * p.nH);
    constexpr bool HAS_LDSB = NT == 9;
    if (HAS_LDSB && !(p.variant & 2))                  // bit 1 of UENC_WATTN_VARIANT: dense bias rows from L2 (A/B)
        hipLaunchKernelGGL((wattn_fwd_kernel<NT, HAS_LDSB>), dim3(grid), dim3(64 * NT), 0, stream, p);
    else
        hipLaunchKernelGGL((wattn_fwd_kernel<NT, false>), dim3(grid), dim3(64 * NT), 0, stream, p);
}
// groups per head: ~one resident workgroup per CU for the 9-wave (12 x 12) case, more for small windows
static int wattn_bwd_groups(int nWinTotal, int nH, int ntiles) {
    const int target = ntiles >= 5 ? 256 : (ntiles >= 3 ? 512 : 1024);
    int G = target / nH;
    if (G < 1) G = 1;
    if (G > nWinTotal) G = nWinTotal;
    return G;
}

template <int NT>
static int launch_bwd(const WAttn& p, float* dtab, int defer_dtable, hipStream_t stream) {
    using Cf = WCfg<NT>;
    constexpr int NK2 = Cf::NK2;
    const int G = wattn_bwd_groups(p.nWinTotal, p.nH, NT);
    constexpr size_t shm = ((size_t)2 * 4 * NK2 * 64 + 2 * NK2 * 4 + 2 * NK2 + 15) / 16 * 16 + NK2 * 4 + 96 * 4 + (size_t)NT * NK2 * 32 + 532 * 4;
    constexpr bool HAS_LDSB = NT == 9;
    constexpr size_t shm_loader = shm + (size_t)2 * NK2 * 64 + NK2 * 4 + NK2 + 16 + 1536 + 16;      // + the fifth image of both stages, the third slot buffer, the row statistics, the mask flags
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)wattn_bwd_kernel<NT, false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
        if (e == hipSuccess && HAS_LDSB) {
            e = hipFuncSetAttribute((const void*)wattn_bwd_kernel<NT, HAS_LDSB, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
            if (e == hipSuccess)
                e = hipFuncSetAttribute((const void*)wattn_bwd_kernel<NT, HAS_LDSB, HAS_LDSB>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm_loader);
            if (e == hipSuccess)
                e = hipFuncSetAttribute((const void*)wattn_bwd_kernel<NT, HAS_LDSB, HAS_LDSB, HAS_LDSB>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm_loader);
        }
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    const int npair = (p.nH + 1) / 2;
    const unsigned grid = (unsigned)((2 * npair * G + 15) / 16 * 16);
    if (HAS_LDSB && !(p.variant & 14) && p.lse != nullptr)      // 12 x 12 windows: LDS bias table, loader wave, saved row statistics
        hipLaunchKernelGGL((wattn_bwd_kernel<NT, HAS_LDSB, HAS_LDSB, HAS_LDSB>), dim3(grid), dim3(64 * (NT + 1)), shm_loader, stream, p, G);
    else if (HAS_LDSB && !(p.variant & 6))             // no statistics given, or bit 3 of UENC_WATTN_VARIANT: recomputed (A/B)
        hipLaunchKernelGGL((wattn_bwd_kernel<NT, HAS_LDSB, HAS_LDSB>), dim3(grid), dim3(64 * (NT + 1)), shm_loader, stream, p, G);
    else if (HAS_LDSB && !(p.variant & 2))             // bit 2 of UENC_WATTN_VARIANT: every wave issues its share of the DMA (A/B)
        hipLaunchKernelGGL((wattn_bwd_kernel<NT, HAS_LDSB, false>), dim3(grid), dim3(64 * NT), shm, stream, p, G);
    else                                               // bit 1: dense bias rows from L2 as well (A/B)
        hipLaunchKernelGGL((wattn_bwd_kernel<NT, false, false>), dim3(grid), dim3(64 * NT), shm, stream, p, G);
    if (!defer_dtable)
        hipLaunchKernelGGL(wattn_dtable_kernel<NT>, dim3((unsigned)(p.nH * NT)), dim3(256), 0, stream, (const float*)p.dtab_ws, dtab, G,
                           p.nH, p.ws);
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

extern "C" int uenc_window_attn_fwd(const void* qkv, const void* qkv_bias, const float* bias_q, void* out, float* lse, int B, int H, int W,
                                    int C, int nH, int ws, int shift, float scale, hipStream_t stream) {
    WAttn p;
    int rc = fill_params(p, qkv, qkv_bias, bias_q, bias_q, B, H, W, C, nH, ws, shift, scale);
    if (rc != UENC_OK) return rc;
    UENC_CHECK_ARG(out != nullptr);
    p.out = (bf16*)out;
    p.lse = lse;
#define CALL(NT) launch_fwd<NT>(p, stream)
    WATTN_DISPATCH(wattn_ntiles(ws), CALL)
#undef CALL
    UENC_LAUNCH_RET();
}

// Workgroup groups per head of uenc_window_attn_bwd (the G of its dense dS partials [nH][G][ntiles][NP * 16]); ntiles = uenc_window_attn_np(ws) / 16.
extern "C" int uenc_window_attn_bwd_groups(int B, int H, int W, int nH, int ws) {
    if (!(B > 0 && H > 0 && W > 0 && nH > 0 && ws >= 1 && ws <= 12)) return 0;
    const long nwin = (long)B * ((H + ws - 1) / ws) * ((W + ws - 1) / ws);
    return wattn_bwd_groups((int)nwin, nH, wattn_ntiles(ws));
}

// Scratch floats for the dense dS partials of uenc_window_attn_bwd.
extern "C" long uenc_window_attn_bwd_ws_floats(int B, int H, int W, int nH, int ws) {
    if (!(B > 0 && H > 0 && W > 0 && nH > 0 && ws >= 1 && ws <= 12)) return 0;
    const long nwin = (long)B * ((H + ws - 1) / ws) * ((W + ws - 1) / ws);
    const int nt = wattn_ntiles(ws);
    return (long)nH * wattn_bwd_groups((int)nwin, nH, nt) * (nt * 16) * (nt * 16);
}

// dgrads: (nH * (2ws-1)^2 + 3C) fp32, overwritten: the gradient of the relative-position table as [head][(2ws-1)^2], then
// the q | k | v (3C) slice of the qkv-bias gradient that flows through padding slots.  dS_ws: scratch of
// uenc_window_attn_bwd_ws_floats() floats.
extern "C" int uenc_window_attn_bwd(const void* qkv, const void* qkv_bias, const float* bias_q, const float* bias_k,
                                    const void* o_saved, const float* lse, const void* d_out, void* dqkv, float* dS_ws, float* dtable, float* dbias_pad,
                                    int B, int H, int W, int C, int nH, int ws, int shift, float scale, int defer_dtable, hipStream_t stream) {
    WAttn p;
    int rc = fill_params(p, qkv, qkv_bias, bias_q, bias_k, B, H, W, C, nH, ws, shift, scale);
    if (rc != UENC_OK) return rc;
    UENC_CHECK_ARG(bias_k && o_saved && d_out && dqkv && dS_ws && dtable && dbias_pad);
    UENC_CHECK_ARG((((uintptr_t)o_saved | (uintptr_t)d_out | (uintptr_t)dqkv | (uintptr_t)dS_ws) & 15) == 0);
    p.o_saved = (const bf16*)o_saved; p.d_out = (const bf16*)d_out; p.dqkv = (bf16*)dqkv; p.dtab_ws = dS_ws;
    p.lse = const_cast<float*>(lse);
    p.dpad = dbias_pad;
#define CALL(NT) { rc = launch_bwd<NT>(p, dtable, defer_dtable, stream); if (rc != UENC_OK) return rc; }
    WATTN_DISPATCH(wattn_ntiles(ws), CALL)
#undef CALL
    UENC_LAUNCH_RET();
}
