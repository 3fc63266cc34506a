/* libuenc_hip.so — C ABI of the MI355X (gfx950) kernels behind the unified-encoder hot path.
 *
 * Drop-in boundary.  The reference is a Python / Detectron2 plug-in whose only foreign-function
 * interface is the pybind11 module `MultiScaleDeformableAttention`
 * (model/modeling/pixel_decoder/ops/src/vision.cpp:18-21, ms_deform_attn.h:25-66).  Entry points
 * `uenc_msdeform_attn_{fwd,bwd}` replace exactly that pair with the same tensor contract.  Every other
 * entry point replaces a group of ATen calls that the reference issues from Python (cited per function);
 * the Python side (`uni-encoder-code_amd/uenc/kernels.py`, ctypes) binds all of them.
 *
 * Conventions
 *   - plain pointers and sizes only: device pointers are `void*` / typed pointers into HBM,
 *     sizes are element counts, strides ("ld*") are in elements; no framework types;
 *   - return 0 = launched, < 0 = invalid argument (shape / alignment / dtype; nothing was launched),
 *     > 0 = hipError_t from the launch;
 *   - the caller owns every buffer (inputs, outputs, scratch); the library allocates nothing, never
 *     synchronises, and launches on the given stream (`uenc_stream_t` = `hipStream_t`, passed as void*);
 *   - stateless and re-entrant, with one opt-in exception: the `uenc_prof_*` launch timers;
 *   - dtype tags: UENC_F32 = 0, UENC_BF16 = 1.  bf16 operands, fp32 accumulation everywhere.
 *   - gfx950 only.
 */
#ifndef UENC_H
#define UENC_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define UENC_F32 0
#define UENC_BF16 1

/* The launch stream.  A HIP stream handle is a pointer; C callers (ctypes, cgo, JNI) pass it as void*.  The library's own
 * sources define UENC_STREAM_T as hipStream_t before including this header, so that hipcc checks every definition in
 * the csrc .hip files against the declaration here (same ABI: one pointer). */
#ifndef UENC_STREAM_T
#define UENC_STREAM_T void*
#endif
typedef UENC_STREAM_T uenc_stream_t;

/* GEMM epilogues */
#define UENC_EPI_NONE 0       /* C = alpha * (A W^T + bias)                                   */
#define UENC_EPI_GELU 1       /* C = gelu(.) (erf form); optional aux_out <- pre-activation    */
#define UENC_EPI_RELU 2       /* C = relu(.)                                                   */
#define UENC_EPI_RESIDUAL 3   /* C = (.) + aux            aux fp32, may alias C                */
#define UENC_EPI_MUL_DGELU 4  /* C = (.) * gelu'(aux)     aux bf16 = saved pre-activation      */
#define UENC_EPI_MUL_DRELU 5  /* C = (.) * (aux > 0)      aux bf16 = saved ReLU output         */

int uenc_version(void);
const char* uenc_arch(void); /* "gfx950" */

/* ---- casts: fp32 master weights -> bf16 MFMA operands (replaces autocast-style .to(bf16)) ---------- */
int uenc_cast_f32_bf16(const float* src, void* dst, long n /* multiple of 8 */, uenc_stream_t stream);
/* Inverted dropout on a bf16 tensor of n elements (n % 8 == 0; in place allowed): out[i] = keep(i) ? in[i] / (1 - p) : 0 with
 * keep(i) = hash(seed, i) >= p * 2^32 (the index hash the attention kernels use for attention-probability dropout).  The mask is a
 * function of (seed, i): the backward calls the same entry on the gradient.  Replaces nn.Dropout between the deformable encoder
 * layer's kernels in training mode (reference pixel_decoder/msdeformattn.py:111-119, 121-142). */
int uenc_dropout_bf16(const void* in, void* out, long n, unsigned seed, float p, uenc_stream_t stream);
int uenc_cast_transpose_f32_bf16(const float* src /* [R][C] */, void* dst /* [C][R] bf16 */, int R, int C, uenc_stream_t stream);

/* batched cast: `table` = n device-resident descriptors {const float* src; bf16* dst; int rows, cols, transpose, tiles_c;
 * long tile_begin;} (40 bytes each; tiles_c = ceil(cols / 64), tile_begin = exclusive prefix sum of 64x64 tile counts);
 * dst is [rows][cols] or, if (transpose & 1), [cols][rows]; transpose >> 4, when non-zero, is the leading dimension of dst in elements
 * (several sources filling row / column blocks of one destination).  One launch refreshes every bf16 weight operand of a model. */
int uenc_cast_multi(const void* table, int n, long total_tiles, uenc_stream_t stream);

/* bilinear resize, align_corners = False, of NC fp32 planes (Hi, Wi) -> (Ho, Wo), Wo % 4 == 0: the final mask upsample
 * F.interpolate(mask_pred_results, size=..., mode="bilinear") of model/oneformer_model.py:255-263 (forward only). */
int uenc_upsample_bilinear(const float* in, float* out, long NC, int Hi, int Wi, int Ho, int Wo, uenc_stream_t stream);

/* attention mask of the masked-attention decoder: mask[r][oy][ox] = bilinear(logits[r], (Ho, Wo))[oy][ox] < 0 (1 = blocked),
 * rows that would be fully blocked are cleared (reference oneformer_transformer_decoder.py:497-505 and :454).
 * logits fp32 [rows][Hi][Wi]; mask u8 [rows][Ho][Wo]. */
int uenc_attn_mask(const float* logits, uint8_t* mask, long rows, int Hi, int Wi, int Ho, int Wo, uenc_stream_t stream);

/* ---- glue of the deformable encoder layer (pixel_decoder/ops/modules/ms_deform_attn.py:91-113) ---------------------
 * out = bf16(a + b), b repeating every `period` elements (n % period == 0, both % 4 == 0): query = src + pos. */
int uenc_add_cast_bf16(const float* a, const float* b, void* out, long n, long period, uenc_stream_t stream);
/* offaw (rows, ld) fp32 = [M][L][P][2] sampling offsets | [M][L*P] attention logits per row (row = image * Lq + query):
 * loc (rows, M, L, P, 2) = ref + off / (W_l, H_l), aw (rows, M, L*P) = softmax(logits).  ref (N|1, Lq, L, 2) fp32
 * (ref_per_image: 1 if it has a batch dimension), shapes (L, 2) int64 device.  L * P <= 16. */
int uenc_msda_prep_fwd(const float* offaw, long ld, const float* ref, int ref_per_image, const int64_t* shapes, float* loc,
                       float* aw, long rows, int Lq, int M, int L, int P, uenc_stream_t stream);
/* doffaw (rows, ld) bf16 <- d(loc), d(aw) and the saved softmax aw. */
int uenc_msda_prep_bwd(const float* dloc, const float* daw, const float* aw, const int64_t* shapes, void* doffaw, long ld,
                       long rows, int Lq, int M, int L, int P, uenc_stream_t stream);
/* out (nseg, 128, cols) fp32 = 128 partial column sums per segment (the caller adds them) of the bf16 matrix x16 over row
 * segments [seg_start[s], seg_start[s+1]) of every image (rows_per_image rows each): per-level sums for the level-embedding
 * gradient.  cols % 8 == 0. */
int uenc_segment_colsum(const void* x16, long ld, int cols, const int64_t* seg_start, int nseg, long rows_per_image, int images,
                        float* out, uenc_stream_t stream);

/* ---- FPN branch of the pixel decoder on token matrices (pixel_decoder/msdeformattn.py:283-304, :343-352) -----------
 * y = GroupNorm(x) [+ bilinear_resize(add_src, align_corners=False)] [ReLU] for x, y (B, HW, C) fp32|bf16 (torch.nn.GroupNorm
 * semantics over (HW x C/G) per image and group).  stats (B, G, 2) = (mean, rstd) is written for the backward; scratch:
 * uenc_groupnorm_tokens_scratch_bytes().  add_src: NULL or fp32 (B, Hs, Ws, C), then HW == H * W.  C % G == 0, (C/G) % 4 == 0,
 * 64 % (C/G) == 0, 256 % (C/4) == 0. */
long uenc_groupnorm_tokens_scratch_bytes(int B, int HW, int C, int G);
int uenc_groupnorm_tokens_fwd(const void* x, int x_dtype, const float* gamma, const float* beta, void* y, int y_dtype,
                              float* stats, void* scratch, const float* add_src, int Hs, int Ws, int H, int W, int B, int HW,
                              int C, int G, float eps, int relu, uenc_stream_t stream);
/* dx (B, HW, C) fp32|bf16; dgamma / dbeta (C) accumulated (may be NULL); relu != 0: the mask is recomputed from x. */
int uenc_groupnorm_tokens_bwd(const void* dy, int dy_dtype, const void* x, int x_dtype, const float* gamma, const float* beta,
                              const float* stats, void* dx, int dx_dtype, float* dgamma, float* dbeta, void* scratch, int B,
                              int HW, int C, int G, int relu, uenc_stream_t stream);
/* adjoint of the bilinear merge: dsrc (B, Hs, Ws, C) fp32 overwritten from dy (B, H, W, C) fp32|bf16, H >= Hs, W >= Ws. */
int uenc_upsample_bilinear_tokens_bwd(const void* dy, int dy_dtype, float* dsrc, int B, int H, int W, int Hs, int Ws, int C,
                                      uenc_stream_t stream);
/* 3x3 / stride 1 / pad 1 convolution as a GEMM (bf16, C % 8 == 0): col (B*H*W, 9*C) with column (ky, kx, c) from
 * in (B, H, W, C), and the adjoint dx (B, H, W, C) from dcol (B*H*W, 9*C). */
int uenc_im2col3x3(const void* in, void* col, int B, int H, int W, int C, uenc_stream_t stream);
int uenc_col2im3x3(const void* dcol, void* dx, int B, int H, int W, int C, uenc_stream_t stream);
/* The same for the 3x3 stride-2 pad-1 convolutions of DiNAT's ConvTokenizer / ConvDownsampler (reference
 * model/modeling/backbone/dinat.py:17-45): col (B * ceil(H/2) * ceil(W/2), 9C) bf16 in (ky, kx, c) order; the adjoint gathers
 * dcol back to dx (B, H, W, C) fp32, every element written once.  C % 8 == 0. */
int uenc_im2col3x3_s2(const void* in, void* col, int B, int H, int W, int C, uenc_stream_t stream);
int uenc_col2im3x3_s2(const void* dcol, float* dx, int B, int H, int W, int C, uenc_stream_t stream);

/* ---- Linear layers ---------------------------------------------------------------------------------
 * C[m][n] = epi(alpha * (sum_k A[m][k] W[n][k] + bias[n])).  A fp32|bf16 [M][K] (lda), W bf16 [N][K] (ldw),
 * C fp32|bf16 [M][N] (ldc).  K % 8 == 0, N % 4 == 0, 16-byte aligned bases.  splitk > 1 or accumulate != 0
 * adds into fp32 C with atomics (EPI_NONE only).
 * Replaces nn.Linear / F.linear / 1x1 Conv2d / einsum("bqc,bchw->bqhw") of
 *   backbone/swin.py:35-41,138-170,335  pixel_decoder/ops/modules/ms_deform_attn.py:103-125
 *   pixel_decoder/msdeformattn.py:126-130,237,266,290  transformer_decoder/oneformer_transformer_decoder.py:183,498-500
 *   and nn.MultiheadAttention's in/out projections (transformer.py:252-253). */
int uenc_gemm_nt(const void* A, int a_dtype, long lda, const void* W, long ldw, void* C, int c_dtype, long ldc,
                 int M, int N, int K, const float* bias, int epilogue, const void* aux, long ldaux,
                 void* aux_out, long ldaux_out, float alpha, int splitk, int accumulate, uenc_stream_t stream);
/* The same with alpha multiplied per SAMPLE: row m uses alpha * sample_scale[m / rows_per_sample] (sample_scale: device fp32).
 * Stochastic depth as an epilogue -- timm DropPath(x) = x * floor(keep + U) / keep per image (reference backbone/swin.py:8, 279, 289):
 * the residual-branch GEMM runs over all images at once, a dropped image's rows come out as the residual alone.  Stored results only. */
int uenc_gemm_nt_scaled(const void* A, int a_dtype, long lda, const void* W, long ldw, void* C, int c_dtype, long ldc,
                        int M, int N, int K, const float* bias, int epilogue, const void* aux, long ldaux,
                        void* aux_out, long ldaux_out, float alpha, const float* sample_scale, int rows_per_sample, uenc_stream_t stream);
/* Linear -> residual add -> LayerNorm with the LayerNorm inside the GEMM's epilogue (pixel_decoder/msdeformattn.py:111-142 at d_model 256,
 * backbone/swin.py:262-295 at C = 192): C (M, N) fp32, ldc == N, receives the pre-norm sum h = A W^T + bias + residual (the LayerNorm
 * backward reads it), y32 / y16 (either may be NULL) the normalised rows as fp32 / bf16, stats (M, 2) = (mean, rstd) (may be NULL).
 * The row must fit one column tile of an LDS-staged kernel: N <= 256, N % 8 == 0, bf16 A, K % 64 == 0, M large enough for the tiled kernels;
 * returns -1 (nothing launched) otherwise: run uenc_gemm_nt + uenc_layernorm_fwd then (same numbers up to fp32 summation order). */
int uenc_gemm_nt_ln(const void* A, int a_dtype, long lda, const void* W, long ldw, void* C, long ldc, int M, int N, int K, const float* bias,
                    const void* residual, long ldres, const float* gamma, const float* beta, float eps, float* y32, void* y16, float* stats,
                    uenc_stream_t stream);
/* Split-K with STORED partial sums (no atomics): split s of `splitk` writes its fp32 partial product to P + s * part_stride
 * (row stride ldp); the caller sums the slices.  splitk must equal uenc_gemm_nt_splits(K, requested) (the number of non-empty
 * k-ranges after rounding to 64).  Replaces the reference's torch.einsum("bqc,bchw->bqhw") backward w.r.t. the mask embedding
 * (model/modeling/transformer_decoder/oneformer_transformer_decoder.py:500) for all prediction heads at once. */
int uenc_gemm_nt_splits(int K, int splitk);
int uenc_gemm_nt_partials(const void* A, int a_dtype, long lda, const void* W, long ldw, float* P, long ldp, long part_stride,
                          int M, int N, int K, float alpha, int splitk, uenc_stream_t stream);

/* `batch` problems of one shape in one launch (problem b: A + b*bsA, W + b*bsW -> C + b*bsC, element strides, byte
 * offsets multiples of 16); no bias / epilogue; split-K and accumulate as above. */
int uenc_gemm_nt_batched(const void* A, int a_dtype, long lda, long bsA, const void* W, long ldw, long bsW, void* C, int c_dtype,
                         long ldc, long bsC, int batch, int M, int N, int K, float alpha, int splitk, int accumulate, uenc_stream_t stream);

/* weight / bias gradient of the same Linear:  dW[n][k] += sum_m dY[m][n] X[m][k];  db[n] += sum_m dY[m][n]
 * (db may be NULL).  dY, X fp32|bf16 row-major; dW, db fp32, accumulated (atomics).  N % 8 == K % 8 == 0.
 * splitm <= 0 lets the library choose the split of the token dimension.  Replaces autograd's
 * mm / sum backward of every Linear above. */
int uenc_gemm_tn(const void* dY, int dy_dtype, long ldy, const void* X, int x_dtype, long ldx, float* dW, long ldw,
                 float* db, int M, int N, int K, int splitm, uenc_stream_t stream);
/* dW += alpha * dY^T X, db += alpha * column sums: the weight gradient of a residual branch whose output was scaled by alpha
 * (stochastic depth, timm DropPath at reference backbone/swin.py:279, 289; the branch's dropped images are left out of M). */
int uenc_gemm_tn_scaled(const void* dY, int dy_dtype, long ldy, const void* X, int x_dtype, long ldx, float* dW, long ldw,
                        float* db, int M, int N, int K, int splitm, float alpha, uenc_stream_t stream);

/* Grouped weight gradients: one launch for many (dY, X, dW, db) problems of the form above (bf16 operands only).
 * table: n descriptors in DEVICE memory, 96 bytes each:
 *   { const void* dY, *X; float* dW, *db; long ldy, ldx, ldw; int M, N, K, tiles_k, mlen, nsplit, item_begin, store; float alpha; int zero; }
 * alpha multiplies the problem's sums (0 is read as 1).
 * M % 64 == 0; mlen (tokens per split, % 64 == 0) * nsplit >= M; tiles_k = ceil(K / tile); item_begin = exclusive prefix
 * sum of ceil(N / tile) * tiles_k * nsplit; total_items = the full sum.  tile = 256 or 128.  Accumulates into dW / db
 * (atomic adds), or, for a descriptor with store != 0 and nsplit == 1, overwrites them with plain stores.
 * flops = 2 * sum(M N K), used by uenc_prof_* only. */
int uenc_gemm_tn_grouped(const void* table, int n, int total_items, int tile, double flops, uenc_stream_t stream);
/* The same for the register-staged kernel (any M, fp32|bf16 operands, 128 x 128 tiles): descriptors of 96 bytes
 *   { const void* dY, *X; float* dW, *db; long ldy, ldx, ldw; int M, N, K, dy_f32, x_f32, tiles_k, mlen, nsplit, item_begin; float alpha; }   (alpha: as above)
 * tiles_k = ceil(K / 128), mlen % 64 == 0, item_begin = exclusive prefix sum of ceil(N / 128) * tiles_k * nsplit. */
int uenc_gemm_tn_grouped_small(const void* table, int n, int total_items, double flops, uenc_stream_t stream);

/* ---- LayerNorm over the last dimension (C % 4 == 0, C <= 6144) --------------------------------------
 * y = LN(x + res) * gamma + beta; optional h_out <- x + res (fp32); optional stats <- (mean, rstd) per row.
 * Replaces nn.LayerNorm and the preceding residual add of swin.py:247,293,334,492,673,
 * msdeformattn.py:128-129,136-137, transformer.py:268-297, oneformer_transformer_decoder.py:66-67,126-127,184-185,496. */
int uenc_layernorm_fwd(const void* x, int x_dtype, const void* res, int res_dtype, float* h_out, const float* gamma,
                       const float* beta, void* y, int y_dtype, float* stats, long M, int C, float eps, void* y16,
                       uenc_stream_t stream);
/* dx = LN'(dy) [+ dres];  dgamma / dbeta accumulated (both NULL to skip).
 * y16 / dx16 (may be NULL): a bf16 copy of y / dx written in the same pass -- the operand the next GEMM reads, so that an
 * fp32 stream needs no separate cast kernel.  part_ws (may be NULL): scratch of 2048 * 2 * C floats; with it the
 * workgroups' dgamma / dbeta partials are stored and summed by a second small kernel instead of added atomically. */
int uenc_layernorm_bwd(const void* dy, int dy_dtype, const void* h, int h_dtype, const float* stats, const float* gamma,
                       const float* dres, void* dx, int dx_dtype, float* dgamma, float* dbeta, long M, int C, void* dx16,
                       float* part_ws, int defer_param_sums, uenc_stream_t stream);
/* Deferred parameter sums.  With part_ws and defer_param_sums != 0 the block partials of dgamma / dbeta stay in part_ws -- [nblk][2][C]
 * floats, nblk = uenc_layernorm_bwd_blocks(M, C) (0: the pass adds its few block sums itself and nothing is deferred) -- and NO reduction is
 * launched; the caller keeps part_ws untouched and later sums the partials of many passes in ONE launch:
 *   table: n descriptors in DEVICE memory, 40 bytes each { const float* part; float* dgamma; float* dbeta; int nblk, C, group_begin, pad; },
 *   group_begin = exclusive prefix sum of ceil(2C / 64), total_groups = the full sum; dgamma / dbeta are accumulated.
 * (A training step leaves ~70 such reductions of a few MB each: 10 us launches for 2 us of work.) */
int uenc_layernorm_bwd_blocks(long M, int C);
int uenc_ln_param_grouped(const void* table, int n, int total_groups, uenc_stream_t stream);
/* PatchMerging's pad-to-even + 2x2 strided gather + concat (order (0,0), (1,0), (0,1), (1,1)) + LayerNorm(4C) as one pass
 * (reference model/modeling/backbone/swin.py:311-334; the 4C -> 2C reduction GEMM follows): x (B, H, W, C) fp32 ->
 * y (B * ceil(H/2) * ceil(W/2), 4C) bf16, stats (rows, 2).  Backward: dy (rows, 4C) bf16 | fp32 -> dx (B, H, W, C) fp32 written
 * completely; dgamma / dbeta (4C) accumulated; part_ws as for uenc_layernorm_bwd. */
int uenc_patch_merge_ln_fwd(const float* x, const float* gamma, const float* beta, void* y, float* stats, int B, int H, int W,
                            int C, float eps, uenc_stream_t stream);
int uenc_patch_merge_ln_bwd(const void* dy, int dy_dtype, const float* x, const float* stats, const float* gamma, float* dx,
                            float* dgamma, float* dbeta, float* part_ws, int B, int H, int W, int C, int defer_param_sums,
                            uenc_stream_t stream);          /* rows M = B * ceil(H/2) * ceil(W/2), width 4C for uenc_layernorm_bwd_blocks */

/* ---- shifted-window attention (head_dim 32, window <= 12) ---------------------------------------------
 * Replaces F.pad -> torch.roll -> window_partition -> WindowAttention core -> window_reverse -> roll -> crop,
 * backbone/swin.py:250-289 around :131-171, shift mask of :414-440.
 * qkv (B,H,W,3C) bf16 = qkv Linear output of the real tokens; qkv_bias (3C) bf16 (value of padding slots);
 * bias_q / bias_k: expanded relative-position bias from uenc_relpos_expand; out (B,H,W,C) bf16. */
int uenc_window_attn_np(int ws); /* padded tokens per window = 16 * ceil(ws*ws / 16) */
int uenc_relpos_expand(const float* table /* ((2ws-1)^2, nH) */, float* bias_q /* (nH,NP,NP) [h][q][key] */,
                       float* bias_k /* (nH,NP,NP) [h][key][q] */, int nH, int ws, uenc_stream_t stream);
/* Grouped form: n descriptors in device memory (8-byte aligned), 40 bytes each
 * { const float* table; float* bias_q; float* bias_k (NULL: not written); int nH, ws, NP, blk_begin; } with NP =
 * uenc_window_attn_np(ws) and blk_begin = exclusive prefix sum of ceil(nH*NP*NP / 2048); total_blocks = the full sum.  One
 * launch refreshes the expanded biases of every window-attention module after an optimizer step (the host keeps them keyed by
 * the table's version: uenc.ops.ParamCache.relpos).  (The window-attention kernels do not read bias_k any more: it may alias
 * bias_q in their calls.) */
int uenc_relpos_expand_grouped(const void* table, int n, int total_blocks, uenc_stream_t stream);
/* lse (B,H,W,nH) fp32 or NULL: the softmax row statistics max + log2(sum) of every real token and head (scores in log2 units), written
 * by the forward; given back to uenc_window_attn_bwd, the 12 x 12 backward computes the probabilities from them instead of repeating
 * the maximum / sum passes (other window sizes, and lse == NULL, recompute). */
int uenc_window_attn_fwd(const void* qkv, const void* qkv_bias, const float* bias_q, void* out, float* lse, int B, int H, int W,
                         int C, int nH, int ws, int shift, float scale, uenc_stream_t stream);
/* dqkv (B,H,W,3C) bf16 written.  dS_ws: scratch of uenc_window_attn_bwd_ws_floats() floats (dense per-workgroup sums of
 * dS, overwritten).  The two parameter gradients are ACCUMULATED (+=, float atomics) straight into the caller's buffers, i.e.
 * into attn.relative_position_bias_table.grad and attn.qkv.bias.grad: dtable ((2ws-1)^2, nH) fp32 in the parameter's own
 * layout; dbias_pad (3C) fp32 = the q | k | v slice of the qkv.bias gradient that flows through padding slots (the zero rows
 * F.pad appends after norm1, swin.py:254, whose q/k/v equal the bias). */
long uenc_window_attn_bwd_ws_floats(int B, int H, int W, int nH, int ws);
int uenc_window_attn_bwd(const void* qkv, const void* qkv_bias, const float* bias_q, const float* bias_k,
                         const void* o_saved, const float* lse /* of uenc_window_attn_fwd, or NULL */, const void* d_out, void* dqkv,
                         float* dS_ws, float* dtable, float* dbias_pad,
                         int B, int H, int W, int C, int nH, int ws, int shift, float scale, int defer_dtable, uenc_stream_t stream);
/* defer_dtable != 0: dqkv and dbias_pad as above, but the relative-position-table gradient is NOT reduced: dS_ws keeps the dense partials
 * [nH][G][ntiles][NP * 16] (G = uenc_window_attn_bwd_groups(...), ntiles = NP / 16) and the caller, keeping dS_ws untouched, reduces the
 * partials of many passes in one launch: table = n descriptors in DEVICE memory, 40 bytes each
 * { const float* wsd; float* dtab; int G, nH, ws, ntiles, blk_begin, pad; }, blk_begin = exclusive prefix sum of nH * ntiles. */
int uenc_window_attn_bwd_groups(int B, int H, int W, int nH, int ws);
int uenc_window_attn_dtable_grouped(const void* table, int n, int total_blocks, uenc_stream_t stream);

/* ---- multi-scale deformable attention: the reference's native op ---------------------------------------
 * ms_deform_attn_forward(value, spatial_shapes, level_start_index, sampling_loc, attn_weight, im2col_step)
 *   (ops/src/ms_deform_attn.h:25-45, kernel ops/src/cuda/ms_deform_im2col_cuda.cuh:242-304):
 * value (B,S,M,D) fp32|bf16, shapes (L,2) int64 (H,W), level_start (L) int64, loc (B,Lq,M,L,P,2) fp32,
 * attn (B,Lq,M,L,P) fp32 -> out (B,Lq,M*D) fp32|bf16.  D in {16,32,64}.  im2col_step is not needed:
 * one launch covers the batch. */
int uenc_msdeform_attn_fwd(const void* value, int v_dtype, const int64_t* shapes, const int64_t* level_start,
                           const float* loc, const float* attn, void* out, int out_dtype, int B, int S, int M, int D,
                           int L, int Lq, int P, uenc_stream_t stream);
/* The same forward for the geometry of the deformable ENCODER (ops/modules/ms_deform_attn.py called from
 * pixel_decoder/msdeformattn.py:115-121: the queries are the pixels of the L maps, Lq == S, level-major), with the value pixels a
 * neighbourhood of queries samples staged once per workgroup in LDS instead of gathered tap by tap from L2.  Same arguments and
 * result plus shapes_host, the HOST copy of `shapes`.  Needs D == 32, bf16 value, L <= 4, L * P <= 16, Lq == S == sum(H_l * W_l);
 * returns -1 (nothing launched) otherwise -- call uenc_msdeform_attn_fwd then.  Any sampling locations are legal: a level whose
 * sampled box does not fit the LDS budget is gathered from memory as in the general kernel. */
int uenc_msdeform_attn_fwd_tiled(const void* value, int v_dtype, const int64_t* shapes, const int64_t* level_start,
                                 const float* loc, const float* attn, void* out, int out_dtype, int B, int S, int M, int D,
                                 int L, int Lq, int P, const int64_t* shapes_host, uenc_stream_t stream);
/* ms_deform_attn_backward (ms_deform_attn.h:47-66, cuh:306-408): grad_value fp32 accumulated (caller zeroes,
 * as the reference's at::zeros_like), grad_loc / grad_attn overwritten.  shapes_host: optional HOST copy of `shapes`,
 * workspace: optional device scratch of uenc_msdeform_attn_bwd_workspace_bytes() bytes (both may be NULL).  With them
 * (D == 32, L <= 4, L*P <= 16) the grad_value contributions are binned per block of value pixels and summed on chip
 * instead of being sent to memory one float atomic per tap and channel.  Same result either way. */
int uenc_msdeform_attn_bwd(const void* value, int v_dtype, const int64_t* shapes, const int64_t* level_start,
                           const float* loc, const float* attn, const void* grad_out, int go_dtype, float* grad_value,
                           float* grad_loc, float* grad_attn, int B, int S, int M, int D, int L, int Lq, int P,
                           const int64_t* shapes_host, void* workspace, long workspace_bytes, uenc_stream_t stream);
long uenc_msdeform_attn_bwd_workspace_bytes(const int64_t* shapes_host, int B, int M, int D, int L, int Lq, int P);
/* Fused form of the same op for callers that own the two projection Linears around it (ops/modules/ms_deform_attn.py:99-125): instead
 * of sampling locations and attention weights the kernels take the projection row they are derived from,
 *   offaw (B * Lq, ld) fp32 = [M][L][P][2] sampling offsets | [M][L * P] attention logits   (sampling_offsets | attention_weights Linear outputs),
 * the reference points ref (B | 1, Lq, L, 2) fp32 (ref_per_image: 1 if ref has a batch dimension) and the level shapes:
 * loc = ref + off / (W_l, H_l), attn = softmax over the L * P logits, both computed inside the kernels.  The backward writes
 * d(offaw) (B * Lq, ld_doffaw) bf16 (all 3 M L P columns) and accumulates grad_value (caller zeroes) -- grad_loc / grad_attn never exist.
 * L * P <= 16, D == 32, ld even; the backward needs shapes_host + workspace (uenc_msdeform_attn_bwd_workspace_bytes() > 0) and
 * returns -1 otherwise.  Same sums as uenc_msda_prep_fwd + uenc_msdeform_attn_fwd (and the backward pair). */
int uenc_msdeform_attn_fused_fwd(const void* value, int v_dtype, const int64_t* shapes, const int64_t* level_start, const float* offaw,
                                 long ld, const float* ref, int ref_per_image, void* out, int out_dtype, int B, int S, int M, int D,
                                 int L, int Lq, int P, uenc_stream_t stream);
/* The fused forward for the encoder's geometry with the value tiles in LDS (uenc_msdeform_attn_fwd_tiled with the locations / weights derived
 * inside the kernel): P == 4, ld % 4 == 0, 16-byte aligned offaw / out; -1 (nothing launched) when not eligible. */
int uenc_msdeform_attn_fused_fwd_tiled(const void* value, int v_dtype, const int64_t* shapes, const int64_t* level_start, const float* offaw,
                                       long ld, const float* ref, int ref_per_image, void* out, int out_dtype, int B, int S, int M, int D,
                                       int L, int Lq, int P, const int64_t* shapes_host, uenc_stream_t stream);
int uenc_msdeform_attn_fused_bwd(const void* value, int v_dtype, const int64_t* shapes, const int64_t* level_start, const float* offaw,
                                 long ld, const float* ref, int ref_per_image, const void* grad_out, int go_dtype, float* grad_value,
                                 void* doffaw, long ld_doffaw, int B, int S, int M, int D, int L, int Lq, int P, const int64_t* shapes_host,
                                 void* workspace, long workspace_bytes, uenc_stream_t stream);

/* ---- decoder multi-head attention core (head_dim 32) --------------------------------------------------------
 * softmax(scale * q k^T [blocked where mask != 0]) v for every nn.MultiheadAttention of the transformer decoder
 * (transformer_decoder/oneformer_transformer_decoder.py:63-67,122-127 with the mask of :504-511; transformer.py:268-297).
 * q (B,Lq,*) / k, v (B,S,*) bf16 with heads interleaved in the row (head h at columns 32h..32h+31); *_bs / *_rs are
 * batch / row strides in elements (multiples of 8).  mask (B,Lq,mask_rs) bytes, shared by all heads, mask_rs a multiple
 * of 4 and >= S, or NULL.  out (B,Lq,*) bf16; lse (B,H,Lq) fp32 log2-domain log-sum-exp (needed by the backward).
 * workspace: uenc_mha_fwd_workspace_floats(...) floats (split-KV partials); NULL when that is 0.
 * dropout_p in [0, 1), seed: dropout on the attention probabilities AFTER the softmax normalisation, as
 * nn.MultiheadAttention(dropout=p) applies it in training mode (transformer.py:249-250, 0.1 in the class transformer): element
 * (b, h, q, key) is kept iff hash(seed, ((b H + h) Lq + q) S + key) >= p 2^32 and scaled by 1 / (1 - p); the backward regenerates
 * the mask from the same (p, seed).  0 = off (inference, and every masked-attention layer: their rate is 0.0, :326-344). */
long uenc_mha_fwd_workspace_floats(int B, int H, int Lq, int S);
int uenc_mha_fwd(const void* q, long q_bs, long q_rs, const void* k, long k_bs, long k_rs, const void* v, long v_bs,
                 long v_rs, const unsigned char* mask, long mask_rs, void* out, long o_bs, long o_rs, float* lse,
                 float* workspace, int B, int H, int Lq, int S, float scale, float dropout_p, unsigned seed, uenc_stream_t stream);
/* dq (B,Lq,*) fp32 ACCUMULATED (caller zeroes); dk, dv (B,S,*) bf16 overwritten for every key and head. */
int uenc_mha_bwd(const void* q, long q_bs, long q_rs, const void* k, long k_bs, long k_rs, const void* v, long v_bs,
                 long v_rs, const unsigned char* mask, long mask_rs, const void* out, long o_bs, long o_rs,
                 const float* lse, const void* dout, long do_bs, long do_rs, float* dq, long dq_bs, long dq_rs, void* dk,
                 long dk_bs, long dk_rs, void* dv, long dv_bs, long dv_rs, int B, int H, int Lq, int S, float scale,
                 float dropout_p, unsigned seed, uenc_stream_t stream);

/* ---- neighbourhood attention 2-D (DiNAT backbone) -------------------------------------------------------------
 * What natten.NeighborhoodAttention2D computes between its qkv and proj Linear layers (reference call site
 * model/modeling/backbone/dinat.py:14, 77-79, 94; NATTEN 0.14.4's natten2dqkrpb + softmax + natten2dav -- the wheel is not
 * part of the reference tree: parity unpinned, restated in oracle/dinat_ref.py).  head_dim is 32.
 * qkv (B, H, W, 3, nH, 32) bf16 = the qkv Linear's output as is; rpb (nH, 2K-1, 2K-1) fp32 or NULL; out (B, H, W, nH, 32)
 * bf16; lse (B, nH, H, W) fp32 natural-log log-sum-exp (NULL for inference).  K odd in 3..13; H, W >= K * dilation (the
 * caller zero-pads smaller inputs first, as NATTEN does); scale = head_dim^-0.5 applied to q.k. */
int uenc_na2d_fwd(const void* qkv, const float* rpb, void* out, float* lse, int B, int H, int W, int nH, int K, int dilation,
                  float scale, uenc_stream_t stream);
/* dqkv (B, H, W, 3, nH, 32) bf16 overwritten completely; drpb (nH, 2K-1, 2K-1) fp32 ACCUMULATED (may be NULL);
 * delta_ws: B * nH * H * W floats of scratch. */
int uenc_na2d_bwd(const void* qkv, const float* rpb, const void* out, const void* dout, const float* lse, void* dqkv, float* drpb,
                  float* delta_ws, int B, int H, int W, int nH, int K, int dilation, float scale, uenc_stream_t stream);

/* ---- segmentation post-processing fused with the mask upsample (inference) ---------------------------------------
 * The reference upsamples the (Q, h, w) mask logits to the padded input size (model/oneformer_model.py:255-263), crops the
 * padding (detectron2 sem_seg_postprocess, :277-279) and then runs semantic_inference (:367-371) / panoptic_inference
 * (:373-434) on the 1.25 GB-per-image result.  These entry points interpolate on the fly from the low-resolution logits
 * (bilinear, align_corners = False to (Hp, Wp); output extent (Ho, Wo) <= (Hp, Wp) = the crop); all tensors fp32 / int32,
 * one image per call.
 *   semantic:        sem (C, Ho, Wo) = sum_q class_prob[q, c] * sigmoid(up(mask_logits[q])); class_prob (Q, Cp), Cp % 32 == 0
 *   panoptic_stats:  ids (Ho, Wo) = argmax_q score[q] * sigmoid(up(m_q)) over queries with score > 0 (first maximum wins);
 *                    counts (3, Q), zeroed by the caller: |ids == q|, |sigmoid(up(m_q)) >= 0.5|, |both| -- what the reference
 *                    reads back with three .item() syncs per query (:399-408)
 *   panoptic_label:  seg (Ho, Wo) = segid[ids] where that query's sigmoid >= 0.5, else 0 (:420-425); segid (Q), 0 = dropped */
int uenc_postproc_semantic(const float* mask_logits, const float* class_prob, float* sem, int Q, int C, int Cp, int hl, int wl,
                           int Hp, int Wp, int Ho, int Wo, uenc_stream_t stream);
int uenc_postproc_panoptic_stats(const float* mask_logits, const float* score, int* ids, int* counts, int Q, int hl, int wl, int Hp,
                                 int Wp, int Ho, int Wo, uenc_stream_t stream);
int uenc_postproc_panoptic_label(const float* mask_logits, const int* ids, const int* segid, int* seg, int Q, int hl, int wl, int Hp,
                                 int Wp, int Ho, int Wo, uenc_stream_t stream);

/* ---- fp32 "exact" arithmetic mode (csrc/exact.hip; UENC_EXACT=1 / uenc.ops.set_exact) -----------------------------------
 * The reference computes in fp32 end to end (AMP off, configs/cityscapes/swin/unified_encoder_cityscapes.yaml:27-28; the pixel
 * decoder forces fp32, pixel_decoder/msdeformattn.py:336,343).  These entry points run the contractions of the path on fp32
 * operands with fp32 accumulation (v_mfma_f32_16x16x4_f32 for the GEMMs, VALU for the two attention cores), so that a
 * deviation from the reference can be split into bf16 rounding (the product's mode) and everything else (must be ~1e-5).
 * A verification mode: same call sites, same epilogues, no performance claim.
 *   gemm_nt_f32:  C = epi(alpha * (A W^T + bias)), all fp32, K % 4 == 0, lda / ldw % 4 == 0; aux / aux_out fp32 (the
 *                 GELU pre-activation, ReLU output or residual of the UENC_EPI_* epilogues); accumulate: C += (EPI_NONE only)
 *   gemm_tn_f32:  dW (N, K) += dY^T X, db (N) += column sums of dY (db may be NULL); dY (M, N), X (M, K) fp32, N, K % 4 == 0
 *   window_attn_f32_{fwd,bwd}: qkv (B, H, W, 3C) fp32, qkv_bias (3C), table ((2ws-1)^2, nH) = relative_position_bias_table,
 *                 out / dout (B, H, W, C); same pad / shift / mask semantics as uenc_window_attn_* (model/modeling/backbone/
 *                 swin.py:250-289, :131-171, :414-440).  bwd: dqkv written, dtable and dbias_pad (3C) accumulated (atomics)
 *   mha_f32_{fwd,bwd}: the decoder's nn.MultiheadAttention cores, tensor contract of uenc_mha_* with fp32 tensors; lse (B, nH, Lq);
 *                 bwd: dq written, dk / dv (zeroed by the caller) accumulated, delta (B, nH, Lq) scratch */
int uenc_gemm_nt_f32(const float* A, long lda, const float* W, long ldw, float* C, long ldc, int M, int N, int K, const float* bias,
                     int epilogue, const float* aux, long ldaux, float* aux_out, long ldaux_out, float alpha, int accumulate, uenc_stream_t stream);
int uenc_gemm_tn_f32(const float* dY, long ldy, const float* X, long ldx, float* dW, long ldw, float* db, int M, int N, int K, uenc_stream_t stream);
int uenc_window_attn_f32_fwd(const float* qkv, const float* qkv_bias, const float* table, float* out, int B, int H, int W, int C, int nH,
                             int ws, int shift, float scale, uenc_stream_t stream);
int uenc_window_attn_f32_bwd(const float* qkv, const float* qkv_bias, const float* table, const float* dout, float* dqkv, float* dtable,
                             float* dbias_pad, int B, int H, int W, int C, int nH, int ws, int shift, float scale, uenc_stream_t stream);
int uenc_mha_f32_fwd(const float* q, long qs0, long qs1, const float* k, long ks0, long ks1, const float* v, long vs0, long vs1,
                     const uint8_t* mask, long mask_row_stride, float* out, long os0, long os1, float* lse, int B, int nH, int Lq, int S,
                     float scale, float dropout_p, unsigned seed, uenc_stream_t stream);
int uenc_mha_f32_bwd(const float* q, long qs0, long qs1, const float* k, long ks0, long ks1, const float* v, long vs0, long vs1,
                     const uint8_t* mask, long mask_row_stride, const float* out, long os0, long os1, const float* lse, const float* dout,
                     long gos0, long gos1, float* dq, long dqs0, long dqs1, float* dk, long dks0, long dks1, float* dv, long dvs0, long dvs1,
                     float* delta, int B, int nH, int Lq, int S, float scale, float dropout_p, unsigned seed, uenc_stream_t stream);

/* ---- launch timers (opt-in, process-global): per-launch HIP events on the launch stream ---------------- */
int uenc_prof_enable(int on); /* also resets */
int uenc_prof_collect(int kind /* 0 gemm_nt (128-tile, skinny), 1 gemm_tn*, 4 gemm_nt256, 5 gemm_nt128 */, double* ms_total, double* flops_total, long* launches);
/* algorithmic bytes (operands read once + results written once) of the recorded launches of `kind` (gemm_nt kinds). */
int uenc_prof_collect_bytes(int kind, double* bytes_total);
/* algorithmic bytes of the NEXT recorded launch whose entry point cannot derive them (the grouped wgrad launches read their
 * problem sizes from a device table the caller built). */
int uenc_prof_next_bytes(double bytes);

#ifdef __cplusplus
}
#endif
#endif /* UENC_H */
