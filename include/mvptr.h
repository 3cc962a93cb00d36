/*
 * mvptr.h — C ABI of libmvptr_hip.so, the MI355X (gfx950) implementation of MVPTR's
 * cross-modal BERT encoder hot path.
 *
 * The reference has no FFI: its "operator interface" for this path is a sequence of PyTorch
 * ops inside oscar/modeling/modeling_vlbert.py and
 * transformers/pytorch_transformers/modeling_bert.py.  Each entry point below names the
 * reference op sequence (file:line, relative to the reference tree) it replaces.
 *
 * Conventions
 *   - all pointers are DEVICE pointers unless stated; tensors are row-major;
 *   - `bf16` buffers are raw uint16 bfloat16; statistics, losses and weight gradients are f32;
 *   - every call enqueues work on `stream` (a hipStream_t passed as void*) and returns
 *     without synchronising the device; no call allocates device memory;
 *   - return 0 on success, a negative mvptr_status otherwise; mvptr_last_error() returns a
 *     thread-local description of the last failure on the calling thread;
 *   - the library keeps no mutable global state (re-entrant, one process per GPU or several
 *     host threads on several devices): kernel selection depends on the call's arguments only; the
 *     environment is not read (the measurement tools use a separate diagnostic build with knobs,
 *     `make -C mvp_pytorch_amd/csrc diag`).
 */
#ifndef MVPTR_H
#define MVPTR_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
  MVPTR_OK = 0,
  MVPTR_BAD_SHAPE = -1,
  MVPTR_BAD_ALIGN = -2,
  MVPTR_WORKSPACE_TOO_SMALL = -3,
  MVPTR_UNSUPPORTED_ARCH = -4,
  MVPTR_HIP_ERROR = -5,
  MVPTR_BAD_ARG = -6
} mvptr_status;

/* mvptr_query `what` codes */
enum { MVPTR_Q_ABI_VERSION = 0, MVPTR_Q_ARCH_OK = 1, MVPTR_Q_NUM_CU = 2 };

#define MVPTR_ABI_VERSION 7

/* GEMM epilogues (see mvptr_gemm_nt) */
typedef enum {
  MVPTR_EPI_BIAS = 0,        /* out0(bf16) = acc + bias                                  */
  MVPTR_EPI_BIAS_GELU = 1,   /* u = acc + bias: out0(u8) = q(gelu_erf'(u)) ; out1(bf16) = gelu_erf(u); q: see below */
  MVPTR_EPI_BIAS_RESID = 2,  /* out0(bf16) = dropout(acc + bias) + aux(bf16)              */
  MVPTR_EPI_GELU_BWD = 3,    /* out0(bf16) = acc * g, g decoded from aux(u8) = the saved q(gelu_erf'(u)) ; colsum -> vec_out f32 */
  MVPTR_EPI_ADD = 4,         /* out0(bf16) = acc + aux(bf16)  (aux may be NULL)           */
  MVPTR_EPI_F32 = 5,         /* out0(f32)  = acc + bias                                   */
  MVPTR_EPI_BIAS_TANH = 6,   /* out0(bf16) = tanh(acc + bias)                             */
  /* ABI 5: the gelu' stash in bf16 (the format of rounds 1-3) for reference-numerics runs and A/B runs of the 8-bit stash */
  MVPTR_EPI_BIAS_GELU_BF16 = 7, /* as BIAS_GELU, out0(bf16) = gelu_erf'(u)                */
  MVPTR_EPI_GELU_BWD_BF16 = 8   /* as GELU_BWD, g = aux(bf16)                             */
} mvptr_epilogue;

/* Dropout descriptor.  One 32-bit hash serves the element pair (2j, 2j+1):
 *   x = ((uint32)j ^ seed_lo) + (uint32)(j >> 32) * 0x9E3779B9;
 *   x ^= x >> 16; x *= 0x7feb352d; x ^= x >> 15; x += seed_hi; x *= 0x846ca68b; x ^= x >> 16;
 *   u16(2j) = x & 0xffff ; u16(2j+1) = x >> 16 ; keep(i) = u16(i) >= thresh16
 * with thresh16 = round(p * 65536); kept values are scaled by 65536/(65536-thresh16).
 * p == 0 (thresh16 == 0) disables dropout.  The element index is op-specific and documented
 * per call.  mvptr_dropout_mask() materialises the same mask for tests. */
typedef struct {
  uint32_t seed_lo;
  uint32_t seed_hi;
  uint32_t thresh16; /* 0 => no dropout */
  uint32_t pad_;
} mvptr_dropout;

int mvptr_query(int what, int64_t* out);
const char* mvptr_last_error(void);

/* C[M,N] = A[M,K] * B[N,K]^T with a fused epilogue; A,B bf16, f32 accumulate (MFMA 16x16x32).
 * Replaces nn.Linear forward (y = x W^T + b): modeling_bert.py:348 (BertSelfOutput.dense),
 * :395 (BertIntermediate.dense + gelu :142-148), :408 (BertOutput.dense), Q/K/V projections
 * modeling_vlbert.py:71-73, and the data-gradient of the same layers (dX = dY * W, with
 * B = W^T as stored by mvptr_cast_pack).
 * lda/ldb/ldc/ld_aux in elements; K % 8 == 0, lda % 8 == 0, ldb % 8 == 0, A/B 16-byte aligned.
 * bias: f32[N] or NULL.  aux: bf16 [M, ld_aux] (residual) or, for EPI_GELU_BWD, u8 [M, ld_aux] (the gelu' stash) or NULL.
 * The gelu' stash (ABI 4; zero point 27 since ABI 6): one byte per element, q = floor(200 g + 27 + d) with the dither d in [0, 1), g = (q - 27) / 200 — gelu_erf' lies in
 * [-0.129, 1.129], 0 and 1 are exact; the dither is the low mantissa byte of u: |error| < 0.005 with zero mean, where the
 * round-to-nearest of ABI 4-5 (|error| <= 0.0025) was a deterministic function of u; out0 of EPI_BIAS_GELU has row stride ldc BYTES, out1 ldc elements.
 * vec_out: f32[N] column sums (EPI_GELU_BWD), accumulated with atomics, or NULL.
 * drop: dropout on (acc + bias) for EPI_BIAS_RESID, element index = m * N + n. */
int mvptr_gemm_nt(const void* A, int64_t lda, const void* B, int64_t ldb, int M, int N, int K,
                  int epilogue, const float* bias, const void* aux, int64_t ld_aux, void* out0,
                  void* out1, int64_t ldc, float* vec_out, const mvptr_dropout* drop,
                  void* stream);

/* Split-K form of mvptr_gemm_nt for few-row, long-K products (the data gradient of the vocabulary decoder,
 * transformers/pytorch_transformers/modeling_bert.py:513-516 backward: [~3 k scored rows, 30 528] x [30 528, 768]):
 * the reduction index is cut into `splits` slices; slab z (f32 [M, ldc], at slabs + z * M * ldc) receives the partial
 * product of slice z with plain stores — the caller adds the slabs in index order (deterministic, no atomics). */
int mvptr_gemm_nt_splitk(const void* A, int64_t lda, const void* B, int64_t ldb, int M, int N, int K, int splits,
                         float* slabs, int64_t ldc, void* stream);

/* LayerNorm folded into the GEMMs on either side of it — north_star's "fused LayerNorm + QKV projection", inference path
 * (BertLayerNorm of BertSelfOutput / BertOutput, transformers/pytorch_transformers/modeling_bert.py:348-352, 407-411, feeding
 * oscar/modeling/modeling_vlbert.py:71-73 and modeling_bert.py:394-397 of the next sub-block).  With x = LN(z) =
 * (z - mean) rstd gamma + beta per row, x W^T + b = rstd (z W'^T - mean c) + d where W' = W with column k scaled by gamma[k],
 * c[n] = sum_k W'[n, k] and d = W beta + b: the consumer GEMM runs on the PRE-LayerNorm rows z and finishes the LayerNorm in
 * its epilogue; the producer GEMM writes z and the partial sums of its row statistics.  No LayerNorm launch, no normalised
 * tensor in HBM.  Forward only: training keeps the LayerNorm kernel, because its OUTPUT is an operand of the weight-gradient
 * GEMMs (dW = dY^T x) and has to exist in memory anyway.
 *   mode MVPTR_LN_FOLD_BIAS   out(bf16) = rstd[m] (A B^T - mean[m] colsum[n]) + bias[n]        A = z, B = W', bias = d
 *   mode MVPTR_LN_FOLD_GELU   out(bf16) = erf-GELU of the same                                  (FFN1; no gelu' stash)
 *   mode MVPTR_LN_RESID_STATS out(bf16) = z = A B^T + bias + r;  r = aux rows (stats NULL) or LN(aux rows) with stats / gamma /
 *                             beta (the residual is itself a pre-LayerNorm tensor); row_partials[m, n / 64, 0..1] = (sum, sum
 *                             of squares) of the stored z over each 64-column strip -> mvptr_ln_stats_finalize
 * stats: f32 [M, 2] = (mean, rstd) per row.  Shapes: N % 256 == 0, K % 64 == 0, K >= 128 (MVPTR_BAD_SHAPE otherwise: the caller
 * keeps the unfused path); row_partials: f32 [M, N / 64, 2]. */
#define MVPTR_LN_FOLD_BIAS 0
#define MVPTR_LN_FOLD_GELU 1
#define MVPTR_LN_RESID_STATS 2
int mvptr_gemm_nt_ln(const void* A, int64_t lda, const void* B, int64_t ldb, int M, int N, int K, int mode, const float* bias,
                     const void* aux, int64_t ld_aux, const float* stats, const float* colsum, const float* gamma,
                     const float* beta, void* out, int64_t ldc, float* row_partials, void* stream);
/* stats[m] = (mean, 1 / sqrt(var + eps)) of row m from the partial sums of MVPTR_LN_RESID_STATS (H = the row length, a multiple
 * of 64; biased variance as BertLayerNorm, modeling_bert.py:242-246); sums in strip order (reproducible). */
int mvptr_ln_stats_finalize(const float* row_partials, int M, int H, float eps, float* stats, void* stream);

/* dW[N,K] (+)= A[M,N]^T * B[M,K] ; A = dY (bf16), B = X (bf16), dW f32 (MFMA 32x32x16,
 * transposed LDS reads, split over M with f32 atomics when accumulate != 0 or splits > 1).
 * Replaces the weight gradient of nn.Linear computed by autograd (addmm backward) for every
 * Linear cited above.  The caller zeroes dW when it wants a fresh gradient.
 * lda % 8 == 0, ldb % 8 == 0 (rows padded so that partial 16-byte chunks at the N/K edge stay
 * inside the row).  colsum: optional f32[N], += column sums of A (the bias gradient of the
 * same layer, computed from the tiles already in LDS). */
int mvptr_gemm_tn(const void* A, int64_t lda, const void* B, int64_t ldb, int M, int N, int K,
                  float* dW, int64_t ldw, float* colsum, void* stream);

/* Several weight gradients in ONE launch (the four nn.Linear of an encoder layer at the end of its
 * backward pass): same operation as mvptr_gemm_tn per problem.  Consecutive problems with equal M
 * share a launch (at most MVPTR_TN_MAX_GROUP each), so the atomic write-out of one problem overlaps
 * the MFMA loop of the next instead of ending every launch with an idle tail. */
#define MVPTR_TN_MAX_GROUP 4
#define MVPTR_TN_STACK_MAX 32
typedef struct {
  const void* A;   /* dY bf16 [M, lda] */
  int64_t lda;
  const void* B;   /* X bf16 [M, ldb] */
  int64_t ldb;
  int M, N, K;
  float* dW;       /* f32 [N, ldw], accumulated */
  int64_t ldw;
  float* colsum;   /* optional f32[N] */
} mvptr_tn_problem;
int mvptr_gemm_tn_multi(const mvptr_tn_problem* problems, int count, void* stream);
/* The same with a caller-provided workspace for the per-split partial tiles ("slabs"): launches of 6 000 .. 24 000 token
 * rows then write every M-split's 256x256 f32 partial tile with plain stores into ws and a second kernel adds a tile's
 * slabs into dW in split order — bitwise reproducible weight gradients, and faster where the f32 atomics of the splits
 * are a quarter to a third of the launch (attention pair at M = 10 917: 91 -> 68 us).  mvptr_gemm_tn_ws_bytes: the
 * bytes such a call would use (0: it writes out with atomics; only shapes and M are read).  ws NULL or too small:
 * atomics.  Groups of one call run one after the other on the stream and share ws.  The ordered reduction adds into dW
 * with plain read-modify-writes: launches that accumulate into the SAME dW must be ordered by a stream (the atomic
 * write-out has no such requirement). */
int mvptr_gemm_tn_multi_ws(const mvptr_tn_problem* problems, int count, void* ws, int64_t ws_bytes, void* stream);
int64_t mvptr_gemm_tn_ws_bytes(const mvptr_tn_problem* problems, int count);

/* Every weight gradient of an encoder STACK in one balanced launch (ABI 5).  The caller keeps the operands (dY, X) of all
 * layers alive until the stack's backward pass is done and hands over the whole list; all problems share M (and rows_dev: a
 * device int32 with the rows actually present, or NULL).  T = the problems' 256 x 256 output tiles, G = one workgroup per
 * CU (max_workgroups > 0: at most that many — a multi-rank job leaves a few CUs to RCCL's kernels): workgroup g sweeps ALL
 * token rows of tile r * G + g in round r < T / G (one contributor per tile, nothing to combine) and the T mod G tiles left
 * over are cut into G equal runs of 32-row steps: every workgroup does the same number of steps (no wave quantisation, at
 * most two partial tiles per workgroup; per-layer launches wrote 66 MB of per-split partial tiles for 35 MB of result at the
 * packed row counts of the text / visual stacks).  Write-out: f32 atomics into dW / colsum (accumulated; full-round tiles have
 * one contributor, so only the left-over tiles' sums depend on arrival order).  Lists longer than MVPTR_TN_STACK_MAX go out
 * as several launches.  Same operation per problem as mvptr_gemm_tn (modeling_bert.py:348,395,408, modeling_vlbert.py:71-73
 * under autograd). */
int mvptr_gemm_tn_stack(const mvptr_tn_problem* problems, int count, const int* rows_dev, int max_workgroups, void* stream);

/* Column sums: out[n] += sum_m X[m,n] (X bf16 [M, ldx]); bias gradients. */
int mvptr_colsum(const void* X, int64_t ldx, int M, int N, float* out, void* stream);

/* Self-attention core, one workgroup per (batch, head), whole sequence resident in LDS.
 * Replaces modeling_vlbert.py:75-100 (transpose_for_scores, QK^T/sqrt(d) + mask, softmax,
 * dropout, PV, merge heads).  qkv: bf16 [B*L, 3*H] (Q | K | V, head-major inside each),
 * mask_add: f32 [B, L] additive mask (0 / -10000) broadcast over heads and queries,
 * ctx: bf16 [B*L, H], lse: f32 [B, heads, L] (row log-sum-exp, saved for backward; may be
 * NULL in inference).  head_dim must be 64, L <= 256; qkv, ctx (and dctx, dqkv in backward) 16-byte aligned (rows move
 * as 16-byte pieces).
 * drop: dropout on the probabilities, element index = ((b*heads + h)*L + q)*Lp + key with
 * Lp = L rounded up to a multiple of 32 (adjacent keys of a query form the hash pairs). */
int mvptr_attention_fwd(const void* qkv, const float* mask_add, void* ctx, float* lse, int B,
                        int L, int heads, const mvptr_dropout* drop, void* stream);

/* Backward of the above: dqkv bf16 [B*L, 3*H] from dctx bf16 [B*L, H].  P is recomputed from
 * qkv and lse; delta = rowsum(dctx*ctx) is computed in-kernel. */
int mvptr_attention_bwd(const void* qkv, const float* mask_add, const void* ctx,
                        const void* dctx, const float* lse, void* dqkv, int B, int L, int heads,
                        const mvptr_dropout* drop, void* stream);

/* The attention probabilities themselves, f32 [B, heads, L, L]: softmax(Q K^T / 8 + mask) of the same bf16 Q | K rows, no
 * dropout.  Replaces the `attention_probs` a layer returns under config.output_attentions (modeling_vlbert.py:85,100-101;
 * collected by CaptionBertEncoder.forward :167-168): an inspection output — the step's kernels never write an L x L tensor. */
int mvptr_attention_probs(const void* qkv, const float* mask_add, float* probs, int B, int L, int heads, void* stream);

/* Row-packed ("unpadded") form of the two calls above: the reference runs every padded slot of
 * every sequence through the encoder (modeling_vlbert.py:430-460 only masks them as keys); here the
 * valid rows of all sequences are packed back to back and sequence b occupies rows
 * [seq_start[b], seq_start[b] + seq_len[b]) of qkv / ctx / dctx / dqkv (device int32 arrays).
 * L is the MAXIMUM length (LDS tile size, lse stride: lse stays f32 [B, heads, L]); mask_add may be
 * NULL (all packed rows are valid keys) or f32 [total_rows].  seq_start == seq_len == NULL is the
 * dense layout.  Results equal the dense call on the valid rows: a key masked with -10000
 * contributes exp(-10000 + ...) = 0 in f32 either way. */
int mvptr_attention_fwd_packed(const void* qkv, const float* mask_add, void* ctx, float* lse,
                               const int* seq_start, const int* seq_len, int B, int L, int heads,
                               const mvptr_dropout* drop, void* stream);
int mvptr_attention_bwd_packed(const void* qkv, const float* mask_add, const void* ctx,
                               const void* dctx, const float* lse, void* dqkv, const int* seq_start,
                               const int* seq_len, int B, int L, int heads,
                               const mvptr_dropout* drop, void* stream);

/* y = LayerNorm(z) * gamma + beta (TF style, eps inside sqrt), optional dropout on y.
 * Replaces BertLayerNorm.forward modeling_bert.py:242-246 (+ nn.Dropout where the reference
 * applies it right after, e.g. BertEmbeddings :275-276, img embedding modeling_vlbert.py:499-503).
 * z: bf16 [M, H]; y: bf16, row r is written to out row  (r / rows_per_group) * group_stride +
 * row_offset + (r % rows_per_group)  (lets the caller write straight into a concatenated
 * sequence buffer; rows_per_group = M, group_stride = 0, row_offset = 0 for the identity).
 * mean/rstd: f32 [M] saved statistics (may be NULL).  H % 8 == 0, H <= 1024.
 * drop element index = r * H + c.  gamma == NULL selects the identity (y = dropout(z)), used for
 * the image embedding when use_img_layernorm is off (modeling_vlbert.py:499-503). */
int mvptr_layernorm_fwd(const void* z, const float* gamma, const float* beta, float eps,
                        void* y, float* mean, float* rstd, int M, int H, int rows_per_group,
                        int group_stride, int row_offset, const mvptr_dropout* drop,
                        void* stream);

/* Backward of LayerNorm (+ the dropout that followed the producing dense layer).
 * dy: bf16, read with the same row remap as y above; z/mean/rstd as saved by forward.
 * y_drop: dropout that was applied to y in forward (or NULL).
 * dz: bf16 [M,H] gradient wrt z.  dd: bf16 [M,H] = dz with `dense_drop` applied (gradient wrt
 * the dense output that was dropped out before the residual add; element index m*H+c);
 * dd may be NULL when dense_drop is NULL/disabled (then dd == dz).
 * dgamma/dbeta/dbias: f32 [H], accumulated into (dbias = colsum(dd); each may be NULL).
 * ws: scratch of mvptr_layernorm_bwd_ws_bytes(M, H) bytes (per-workgroup column partials, summed
 * by a second tiny kernel: one contended atomic row would serialise at ~0.09 TB/s on MI355X). */
int64_t mvptr_layernorm_bwd_ws_bytes(int M, int H);
int mvptr_layernorm_bwd(const void* dy, const void* z, const float* mean, const float* rstd,
                        const float* gamma, void* dz, void* dd, float* dgamma, float* dbeta,
                        float* dbias, int M, int H, int rows_per_group, int group_stride,
                        int row_offset, const mvptr_dropout* y_drop,
                        const mvptr_dropout* dense_drop, void* ws, int64_t ws_bytes, void* stream);

/* Embedding gather + add: z[r, :] = word[ids[r]] + pos[pos_ids[r]] + type[type_ids[r]] (bf16 out).
 * Replaces BertEmbeddings.forward modeling_bert.py:268-273 (LayerNorm + dropout follow through
 * mvptr_layernorm_fwd).  Tables are f32 master weights; ids are int64. */
int mvptr_embed_fwd(const int64_t* ids, const int64_t* pos_ids, const int64_t* type_ids,
                    const float* word, const float* pos, const float* type, void* z, int rows,
                    int H, int64_t vocab, int64_t npos, int64_t ntype, void* stream);

/* Backward: scatter-add dz (bf16 [rows,H]) into the three f32 gradient tables (atomics).
 * Replaces embedding_dense_backward of the three nn.Embedding tables. */
int mvptr_embed_bwd(const int64_t* ids, const int64_t* pos_ids, const int64_t* type_ids,
                    const void* dz, float* dword, float* dpos, float* dtype, int rows, int H,
                    void* stream);

/* f32 -> bf16 cast with optional K padding and optional transposed copy.
 * src f32 [rows, cols] (row stride ld_src) -> dst bf16 [rows, ld_dst] (cols..ld_dst zero filled)
 * and, if dst_t != NULL, dst_t bf16 [cols, ld_dst_t] = src^T written at column offset col_off_t.
 * Used for (a) bf16 working copies of the f32 master weights, packed (Q|K|V rows) and
 * transposed for the data-gradient GEMMs, (b) the 2054-d region features
 * (modeling_vlbert.py:498 input) cast to bf16 with K padded to a multiple of 8. */
int mvptr_cast_pack(const float* src, int64_t ld_src, int rows, int cols, void* dst,
                    int64_t ld_dst, void* dst_t, int64_t ld_dst_t, int col_off_t, void* stream);

/* Many mvptr_cast_pack jobs in one launch (all working copies of an encoder's weights after an
 * optimizer step).  `tasks` and `tile_base` live in DEVICE memory: tile_base[i] = first 32x32 tile
 * of task i, tile_base[n_tasks] = total_tiles; tiles_x = ceil(max(cols, ld_dst) / 32).
 * dst_f32 (optional) receives an unconverted f32 copy [rows, cols] (packed Q|K|V bias). */
typedef struct {
  const float* src;
  int64_t ld_src;
  int rows, cols;
  void* dst;        /* bf16 [rows, ld_dst] or NULL */
  int64_t ld_dst;
  void* dst_t;      /* bf16 [cols, ld_dst_t] or NULL */
  int64_t ld_dst_t;
  int col_off_t;
  int tiles_x;
  float* dst_f32;   /* f32 [rows, cols] or NULL */
} mvptr_cast_task;
int mvptr_cast_multi(const mvptr_cast_task* tasks, const int* tile_base, int n_tasks,
                     int total_tiles, void* stream);

/* bf16 -> f32 strided copy (grad hand-back / outputs) */
int mvptr_cast_f32(const void* src, int64_t ld_src, int rows, int cols, float* dst,
                   int64_t ld_dst, void* stream);

/* Cross-entropy over f32 logits [M, ld] with int64 labels (-1 = ignore):
 * loss_row[m] = lse - logit[label] (0 when ignored), lse_row[m] saved.
 * Replaces CrossEntropyLoss(ignore_index=-1) modeling_vlbert.py:1228-1251 on the MLM /
 * masked-concept logits.  bwd writes dlogits bf16 [M, ld_d] = (softmax - onehot) * scale[0]
 * where scale is a device f32 scalar (upstream gradient / number of valid rows). */
int mvptr_ce_fwd(const float* logits, int64_t ld, const int64_t* labels, float* loss_row,
                 float* lse_row, int M, int V, void* stream);
int mvptr_ce_bwd(const float* logits, int64_t ld, const int64_t* labels, const float* lse_row,
                 const float* scale, void* dlogits, int64_t ld_d, int M, int V, int Vpad,
                 void* stream);

/* Vocabulary decoder + CrossEntropyLoss(ignore_index < 0) WITHOUT the [M, V] f32 logits in HBM:
 * replaces `prediction_scores = decoder(h) + bias` followed by the masked-LM loss
 * (transformers/pytorch_transformers/modeling_bert.py:513-516; oscar/modeling/modeling_vlbert.py:1112-1125,
 * 1245-1249) for callers that only need the loss.  h: bf16 [M, ldh] (K columns), W: bf16 [V, ldw],
 * bias: f32 [V] or NULL, labels: int64 [M] (rows with a label outside [0, V) are not scored).
 * Forward: the logits exist tile by tile in the GEMM epilogue only; `part` (f32 [M, ceil(V/64), 2]) takes
 * the per-64-column (max, sum exp) partials, `lab_logit` (f32 [M]) the logit at the label; a second
 * small kernel writes lse_row[m] and loss_row[m] = lse - logit[label] (0 for unscored rows). */
int mvptr_decoder_ce_fwd(const void* h, int64_t ldh, const void* W, int64_t ldw, const float* bias,
                         const int64_t* labels, int M, int V, int K, float* part, float* lab_logit,
                         float* loss_row, float* lse_row, void* stream);
/* Backward: the logits are recomputed by the same GEMM; its epilogue writes
 * dlogits[m, n] = (exp(logit - lse_row[m]) - [n == label[m]]) * scale[0] as bf16 [M, ld_d] with columns
 * V..Vpad-1 zero (0 everywhere for unscored rows): the operand of the data / weight gradient GEMMs. */
int mvptr_decoder_ce_bwd(const void* h, int64_t ldh, const void* W, int64_t ldw, const float* bias,
                         const int64_t* labels, const float* lse_row, const float* scale, int M, int V,
                         int K, void* dlogits, int64_t ld_d, int Vpad, void* stream);

/* Fused multi-tensor AdamW step with the numerics of
 * transformers/pytorch_transformers/optimization.py:131-187: m = b1 m + (1-b1) g;
 * v = b2 v + (1-b2) g^2; p -= step_size * m / (sqrt(v) + eps); p *= decay  (decay = 1 - lr*wd,
 * applied after the Adam update; step_size carries lr and the bias correction).
 * `table` is a DEVICE array of n_tensors descriptors; chunk_tensor / chunk_offset (device, one
 * entry per workgroup) map each chunk of `chunk_elems` elements to its tensor.
 * grad_scale: optional DEVICE f32 scalar multiplied into every gradient element (the clip coefficient of
 * mvptr_clip_coef: second pass of the global-norm clip, oscar/run_pretrain_ml.py:636-640); NULL = 1. */
typedef struct {
  float* p;
  const float* g;
  float* m;
  float* v;
  int64_t n;
  float step_size;
  float decay;
} mvptr_adamw_tensor;
int mvptr_adamw_multi(const mvptr_adamw_tensor* table, const int32_t* chunk_tensor,
                      const int64_t* chunk_offset, int n_chunks, int chunk_elems, float beta1,
                      float beta2, float eps, const float* grad_scale, void* stream);

/* The same update for tensors that have bf16 WORKING COPIES (the operands of the forward and data-gradient
 * GEMMs): each 64 x 64 tile of the tensor, viewed as [rows, cols], is updated and written out in one pass as
 * dst bf16 [rows, ld_dst] (columns cols..ld_dst-1 zero-filled), dst_t bf16 [cols, ld_dst_t] at column offset
 * col_off_t (transposed; query | key | value land side by side in one [H, 3H] matrix) and dst_f32 (f32 copy,
 * the packed Q|K|V bias) — each optional.  Replaces optimizer.step() + the `.to(bf16)` / `.t().contiguous()`
 * weight preparation that follows it (what mvptr_cast_multi does as a separate pass).  tile_base[i] = first
 * tile of tensor i (DEVICE, n_tensors + 1 entries), tiles per tensor = ceil(rows/64) * ceil(max(cols, ld_dst)/64). */
typedef struct {
  float* p;
  const float* g;
  float* m;
  float* v;
  int rows, cols;
  float step_size;
  float decay;
  void* dst;
  int64_t ld_dst;
  void* dst_t;
  int64_t ld_dst_t;
  int col_off_t;
  int pad_;
  float* dst_f32;
} mvptr_adamw_mirror_tensor;
int mvptr_adamw_mirror_multi(const mvptr_adamw_mirror_tensor* table, const int* tile_base, int n_tensors,
                             int total_tiles, float beta1, float beta2, float eps, const float* grad_scale,
                             void* stream);

/* Global gradient-norm clip (torch.nn.utils.clip_grad_norm_, norm_type 2; oscar/run_pretrain_ml.py:636-640,
 * DeepSpeed recipe gradient_clipping 10.0 oscar/tmp_config.json) in two passes over flat f32 gradient buffers:
 * mvptr_sumsq_partial writes one partial sum of squares per 16 384 elements (mvptr_sumsq_partials(n) of them, at
 * `partials`; call once per buffer with consecutive slices of one partial array); mvptr_clip_coef adds the
 * partials in index order (bitwise reproducible) and writes norm_out[0] = sqrt(sum),
 * coef_out[0] = min(1, max_norm / (norm + 1e-6)), which the AdamW kernels take as grad_scale. */
int64_t mvptr_sumsq_partials(int64_t n);
int mvptr_sumsq_partial(const float* x, int64_t n, float* partials, void* stream);
int mvptr_clip_coef(const float* partials, int n, float max_norm, float* norm_out, float* coef_out, void* stream);

/* ---- B-row heads in f32 (heads.hip) ---------------------------------------------------------------------
 * C[M,N] = act(alpha * op(A) op(B) + bias) with exact f32 FMA accumulation: the pooler
 * tanh(dense(h[:,0])) (transformers/pytorch_transformers/modeling_bert.py:468-474), the 2-way image-text-matching
 * head `seq_relationship` (oscar/modeling/modeling_vlbert.py:975-979), the CLIP-style global projections
 * `txt_out[:,0] @ txt_proj` (:525-526), the similarity matrix `global_txt @ global_img.t()` (:527) and the
 * gradients of all four.  op(X) = X or X^T (trans_x); A / B are f32 or bf16 (x_bf16) with unit column
 * stride; a_rows / b_rows (optional, int32) gather A's / B's STORED rows — the [CLS] rows of a padded or
 * row-packed sequence buffer are read in place; act: 0 none, 1 tanh; accumulate != 0: C += (a gradient arena). */
int mvptr_sgemm_small(const void* A, int64_t lda, int a_bf16, int trans_a, const int32_t* a_rows, const void* B,
                      int64_t ldb, int b_bf16, int trans_b, const int32_t* b_rows, int M, int N, int K, float alpha,
                      const float* bias, int act, int accumulate, float* C, int64_t ldc, void* stream);

/* Mean cross entropy over M rows of V <= 64 classes (the 2-way image-text-matching loss,
 * CrossEntropyLoss(ignore_index=-1) oscar/modeling/modeling_vlbert.py:1247-1251; rows whose label is outside
 * [0, V) are ignored): loss[0] = mean; dlogits (optional, f32 [M, V]) = d loss / d logits. */
int mvptr_ce_mean_small(const float* logits, int64_t ld, const int64_t* labels, int M, int V, float* loss,
                        float* dlogits, void* stream);

/* Head glue (round 4).  mvptr_masked_mean: out[0] = sum(loss_row[0..M)) / max(count, 1), out[1] = max(count, 1), count = rows
 * with label >= 0 — the mean of the fused decoder + cross-entropy's per-row losses (CrossEntropyLoss(ignore_index=-1),
 * oscar/modeling/modeling_vlbert.py:1247-1251; unscored rows carry loss_row = 0); one workgroup, fixed reduction tree.
 * mvptr_dgelu_mul: out[m, n] = bf16(dy[m, n] * gelu'(u)[m, n]) for n < N, 0 for N <= n < Npad, gelu' decoded from the 8-bit
 * stash of MVPTR_EPI_BIAS_GELU (the GELU backward of a head transform, modeling_bert.py:142-148 under autograd). */
int mvptr_masked_mean(const float* loss_row, const int64_t* labels, int M, float* out, void* stream);
int mvptr_dgelu_mul(const void* dy, int64_t ld_dy, const void* stash, int64_t ld_s, int stash_bf16, void* out, int64_t ld_o, int M, int N,
                    int Npad, void* stream);

/* g = y / max(||y||_2, eps) per row, inv_norm[r] = 1 / max(||y||, eps): F.normalize(p=2, dim=-1) of
 * oscar/modeling/modeling_vlbert.py:525-526; backward dy = (dg - g (g . dg)) * inv_norm. */
int mvptr_l2norm_fwd(const float* y, float* g, float* inv_norm, int rows, int H, float eps, void* stream);
int mvptr_l2norm_bwd(const float* g, const float* inv_norm, const float* dg, float* dy, int rows, int H, void* stream);

/* Symmetric contrastive loss of oscar/modeling/modeling_vlbert.py:1238-1241 over sim f32 [n, n] (row = text,
 * column = image): logits = sim * exp(logit_scale[0]); loss[0] = (mean_i CE(logits[i,:], i) + mean_j
 * CE(logits[:,j], j)) / 2.  lse: f32 [2n] (row then column log-sum-exp, kept for backward); parts: f32 [2n]
 * scratch.  Backward: dsim f32 [n, n] = d loss / d sim * gloss[0]; dlogit_scale[0] += d loss / d logit_scale
 * * gloss[0] (optional); parts: f32 [n] scratch. */
int mvptr_clip_ce_fwd(const float* sim, int n, int64_t ld, const float* logit_scale, float* lse, float* parts,
                      float* loss, void* stream);
int mvptr_clip_ce_bwd(const float* sim, int n, int64_t ld, const float* logit_scale, const float* lse,
                      const float* gloss, float* dsim, float* parts, float* dlogit_scale, void* stream);

/* In-batch hard negatives, hn_mod = 'hard' (oscar/modeling/modeling_vlbert.py:529-566): hard_img[i] = argmax_j (sim - 2 I)[i, j]
 * (per text the most similar other image), hard_txt[j] = argmax_i (sim - 2 I)[i, j]; the lowest index wins a tie.  With
 * perm (int64 [n], the caller's torch.randperm — the draw stays torch's): hard_txt_full = [perm[:n/2] ; hard_txt[perm[n/2:]]],
 * hard_img_full = [hard_img[perm[:n/2]] ; perm[n/2:]] (vl:544-566) and, when given, sel_txt / sel_img int64 [2n] =
 * [0..n) ++ hard_*_full (the `sel` vectors of the joint + hard-negative mvptr_pack_maps call).  All int64 device arrays. */
int mvptr_hard_negative_mine(const float* sim, int n, int64_t ld, const int64_t* perm, int64_t* hard_img, int64_t* hard_txt,
                             int64_t* hard_txt_full, int64_t* hard_img_full, int64_t* sel_txt, int64_t* sel_img, void* stream);

/* instance_bce_with_logits (oscar/modeling/modeling_vlbert.py:878-883; the VQA loss, loss_type 'bce'):
 * loss[0] = sum_{r,c} (max(x,0) - x y + log(1 + exp(-|x|))) / rows = binary_cross_entropy_with_logits(mean) * cols;
 * dlogits (optional, f32 [rows, cols]) = (sigmoid(x) - y) / rows.  logits / labels f32 contiguous [rows, cols];
 * parts: f32 [n_parts] scratch (one partial sum per workgroup, added in index order: reproducible). */
int mvptr_bce_logits(const float* logits, const float* labels, int rows, int cols, float* loss, float* dlogits,
                     float* parts, int n_parts, void* stream);

/* ---- rows between the stacks (rows.hip) ------------------------------------------------------------------
 * Row gather / scatter-add over bf16 [rows, H] buffers: out[i,:] = src[idx[i],:] (idx < 0: zero row) and
 * dst[idx[i],:] += src[i,:] (src bf16, or f32 when src_f32; rows may repeat: f32 atomics into an f32 destination
 * when dst_f32 — order-independent to 2^-24, the caller rounds once — else packed bf16 atomics).  With a
 * second buffer (src2 / dst2 != NULL) indices >= split address row idx - split of it: two packed stack outputs
 * used as one source.  Replace the index_select / masked_select / torch.cat chains that move rows between the
 * stacks and into the heads (oscar/modeling/modeling_vlbert.py:519,544-552,586-590,1231-1234,1245;
 * modeling_bert.py:471) and the index_add / zero-fill-and-copy kernels autograd derives for them.
 * H % 8 == 0 (gather), H % 2 == 0 (scatter). */
int mvptr_gather_rows(const void* src, int64_t ld_src, const void* src2, int64_t ld_src2, int split,
                      const int32_t* idx, void* out, int64_t ld_out, int n, int H, void* stream);
int mvptr_scatter_add_rows(const void* src, int64_t ld_src, int src_f32, const int32_t* idx, void* dst,
                           int64_t ld_dst, void* dst2, int64_t ld_dst2, int split, int dst_f32, int n, int H,
                           void* stream);

/* Gradient of a multi-tap row gather in one pass over the DESTINATION rows (the backward of several mvptr_gather_rows
 * calls on one source pair, engine.MultiTapFn): dst[r,:] (bf16) = sum over taps k and positions j with taps[k].idx[j] == r
 * of taps[k].g[j,:] (bf16, or f32 when g_f32), accumulated in f32, rounded once; rows nobody tapped become zero rows, so
 * the destination needs no initialisation.  Indices >= rows address row idx - rows of dst2 ([rows2, H]; NULL with
 * rows2 = 0), negative indices contribute nothing.  The taps are inverted into `work` (>= 2 (rows + rows2 + 1) + sum of n
 * int32, 8-byte aligned, contents irrelevant: a linked list per destination row); a row's contributions are summed in
 * ascending (tap, position) order when there are <= 64 of them (result independent of arrival order), in arrival order
 * beyond.
 * Replaces the zero-fill + index_add + cast kernels autograd derives for oscar/modeling/modeling_vlbert.py:519,
 * 544-552,586-590,1231-1234,1245 and modeling_bert.py:471.  H % 4 == 0, H <= 2048, n < 2^24 per tap. */
#define MVPTR_TAP_MAX 12
typedef struct mvptr_tap {
  const void* g;        /* [n, H] rows, leading dimension ld_g elements */
  int64_t ld_g;
  const int32_t* idx;   /* [n] destination rows */
  int n;
  int g_f32;            /* 1: g is f32, 0: bf16 */
} mvptr_tap;
int mvptr_tap_rows_bwd(const mvptr_tap* taps, int ntaps, void* dst, int64_t ld_dst, int rows, void* dst2, int64_t ld_dst2,
                       int rows2, int H, int32_t* work, int64_t work_elems, void* stream);

/* Scored rows of a masked-LM head in one launch: the slots (b, l) of labels[B, L] with label > -1, ascending, as
 * out_labels[k] = the label and out_rows[k] = pos[b * ld_pos + l] (int32 row map of the packed buffer; NULL: the flat slot
 * index).  Exactly n_out entries are written: a shortfall is padded with label -1 / row -1 (ignored by the loss); MORE
 * scored slots than n_out is a caller bug that drops rows from the loss — the kernel stays in bounds and reports it through the
 * DEVICE ERROR WORD `err` (ABI 7; int64 [4] device memory owned by the caller, zero = no error: err[0] = the first
 * MVPTR_DEV_ERR_* code raised since the caller last cleared it, err[1..2] = the two counts; NULL: a printf only).  The host reads
 * the word whenever it next reads anything back from the device and raises there — the reference raises catchably at this
 * point (oscar/modeling/modeling_vlbert.py:435,542), a trap (ABI 5-6) took the whole process down.
 * Replaces the masked_select chains of oscar/modeling/modeling_vlbert.py:1231-1234,1245. */
enum { MVPTR_DEV_ERR_SCORED_ROWS = 1,   /* mvptr_compact_scored: more scored slots than output slots */
       MVPTR_DEV_ERR_PHRASES = 2,       /* host-side check queued by the model: a sample has more phrases than config.max_phrases */
       MVPTR_DEV_ERR_FEW_REGIONS = 3 }; /* config.wra_strict: an image with phrases has fewer than 3 regions (topk(3), vl:1547) */
int mvptr_compact_scored(const int64_t* labels, const int32_t* pos, int64_t ld_pos, int B, int L, int n_out,
                         int64_t* out_labels, int32_t* out_rows, int64_t* err, void* stream);

/* Index maps of a row-packed pass, built on the device from additive attention masks (valid slot <=> 0).
 * Output sequence s (0 <= s < n_seq) is the concatenation of nseg (1 or 2) segments; segment k covers the slots
 * [col0, col0 + len) of mask row sel[s] (s when sel == NULL) and names the SOURCE row of each valid slot:
 * pos[sel * ld_pos + col] when pos != NULL (a row of an already packed buffer: the joint sequence =
 * text rows of the packed text output + region rows of the packed visual output, hard negatives through sel,
 * oscar/modeling/modeling_vlbert.py:544-552,586-590), else sel * src_seq_stride + col (a row of a padded
 * [*, src_seq_stride, H] buffer: the uni-modal stacks, vl:430-460); src_base is added (offset of a second buffer).
 * Outputs: pos_out int32 [n_seq, sum len] = packed row of every slot (-1: padded slot), idx_out int32
 * [>= total valid] = source row of every packed row (the index vector of mvptr_gather_rows), seq_start /
 * seq_len int32 [n_seq] (mvptr_layer_desc), counts int64 [2] = {total valid rows, longest sequence}. */
typedef struct {
  const float* mask;
  int64_t ld_mask;
  const int64_t* sel;
  int col0, len;
  const int32_t* pos;
  int64_t ld_pos;
  int64_t src_seq_stride;
  int64_t src_base;
} mvptr_pack_seg;
int mvptr_pack_maps(const mvptr_pack_seg* segs, int nseg, int n_seq, int32_t* pos_out, int32_t* idx_out,
                    int32_t* seq_start, int32_t* seq_len, int64_t* counts, void* stream);

/* Host-provided counts against the device's: counts_a / counts_b are the `counts` outputs of two mvptr_pack_maps calls (int64 [2]:
 * rows, longest); a value that differs from the host's number traps the kernel (the process aborts with a message).  This is the
 * one device-side check that cannot report through an error word and carry on: every buffer, grid and LDS tile of the kernels
 * queued behind it was sized from the host's numbers, so continuing would index out of bounds.  (Callers that want a catchable
 * error check on the host with one read-back instead: BiBertImgModel.verify_host_counts.) */
int mvptr_check_counts(const int64_t* counts_a, const int64_t* counts_b, int64_t rows_a, int64_t lmax_a, int64_t rows_b, int64_t lmax_b,
                       void* stream);

/* Word-region alignment loss (phrase_mod == 'sample'), replaces oscar/modeling/modeling_vlbert.py:1285-1300 with
 * get_pos_neg_sims :1553-1596 and t2i_sim :1543-1550 on rows already gathered from the joint output:
 * txt bf16 [n, Pw, H] (row k of sample i = its k-th phrase row, rows beyond phrase_index[i,1] - phrase_index[i,0] are
 * ignored), reg bf16 [n, Rw, H] (region rows, img_index[i,1] - img_index[i,0] of them valid).  pos_pick / neg_pick
 * int64 [n, Pw] in 0..2 are the reference's randint(0, 3) draws (which of the three most similar regions a phrase
 * takes, vl:1547-1549), neg_img int64 [n] the other image each sample is compared with (vl:1572-1573).
 * loss f32 [1] = mean over the samples with phrases of max(neg - pos + 0.2, 0).  The remaining outputs are what the
 * backward pass reads: hinge f32 [n], coef f32 [n], cnt int32 [n, 2], sel int32 [n, Pw, 2], sval f32 [n, Pw, 2],
 * inv_p f32 [n, Pw], inv_r f32 [n, Rw].  mvptr_wra_bwd: gout = d/d loss (one f32 in device memory) ->
 * d_txt bf16 [n, Pw, H], d_reg bf16 [n, Rw, H] (every row written; zero where nothing flows).
 * mvptr_wra_rows builds the row vectors the gathers use from the packed-row map `pos` int32 [>= n, Lj] of
 * mvptr_pack_maps: rows_p int32 [n, Pw], rows_r int32 [n, Rw], -1 beyond a sample's counts.
 * H even and <= 1024; Pw <= 512; Pw * H * 2 + 8 * Pw * Rw bytes of LDS must fit (150 KB). */
int mvptr_wra_rows(const int32_t* pos, int Lj, const int64_t* phrase_index, const int64_t* img_index, int n, int Pw,
                   int Rw, int32_t* rows_p, int32_t* rows_r, void* stream);
int mvptr_wra_fwd(const void* txt, const void* reg, const int64_t* phrase_index, const int64_t* img_index,
                  const int64_t* pos_pick, const int64_t* neg_pick, const int64_t* neg_img, int n, int Pw, int Rw, int H,
                  float* loss, float* hinge, float* coef, int32_t* cnt, int32_t* sel, float* sval, float* inv_p,
                  float* inv_r, void* stream);
int mvptr_wra_bwd(const void* txt, const void* reg, const int64_t* neg_img, int n, int Pw, int Rw, int H,
                  const int32_t* cnt, const int32_t* sel, const float* sval, const float* inv_p, const float* inv_r,
                  const float* coef, const float* gout, void* d_txt, void* d_reg, void* stream);

/* Input pipeline: region features of n_samples TSV rows, still base64 text, ->
 * out_f32 [n_samples, R, D] and/or out_bf16 [n_samples * R, ld_bf16] (columns D..ld_bf16-1 zero: the
 * K-padded operand of the region-embedding GEMM).  Replaces get_img_feature
 * oscar/oscar_datasets_ml/oscar_tsv4.py:696-724 (np.frombuffer(base64.b64decode(arr[-1]), float32)
 * .reshape(num_boxes, img_feature_dim)), the truncation to max_img_seq_length rows and zero padding
 * of __getitem__ :332-352, and data_process' images.to(dtype) run_pretrain_ml.py:501-504.
 * text: device bytes; sample s occupies text[offsets[s] .. offsets[s] + n_chars[s]), offsets 16-byte
 * aligned and the buffer readable up to the next multiple of 16 after every sample.  offsets, n_chars
 * (int64) and num_boxes (int32) are device arrays.  err_flag (device int32, zero it first) gets
 * bit 0 if a sample's text is shorter than num_boxes x D floats need (or misaligned), bit 1 for a
 * character outside the RFC 4648 alphabet inside the kept rows (base64.b64decode would skip it). */
int mvptr_b64_decode_features(const void* text, const int64_t* offsets, const int64_t* n_chars,
                              const int32_t* num_boxes, int n_samples, int R, int D, float* out_f32,
                              void* out_bf16, int64_t ld_bf16, int32_t* err_flag, void* stream);

/* Materialise the dropout keep-mask (1/0 bytes) for n elements — test support. */
int mvptr_dropout_mask(const mvptr_dropout* drop, int64_t n, uint8_t* keep, void* stream);
/* Dropout salt (ABI 7).  Dropout seeds are launch ARGUMENTS (mvptr_dropout, mvptr_layer_desc.seed): a captured HIP graph replays
 * them as captured, i.e. every replay would drop the same elements.  `word` (device uint32, owned by the caller, registered for
 * the CURRENT device; NULL unregisters) is read by every dropout-applying kernel at its start and mixed into its seeds
 * (seed_lo ^= w * 0x9E3779B9, seed_hi += w * 0x85EBCA6B): a captured training step increments the word once per replay and gets
 * fresh masks.  A word of 0 — or none — leaves the documented masks untouched (eager steps).  Host call, no stream. */
int mvptr_set_dropout_salt(const uint32_t* word);

/* One BERT encoder layer, forward and backward, composed from the kernels above.
 * Replaces CaptionBertLayer.forward modeling_vlbert.py:191-199 (attention :63-103,
 * BertSelfOutput modeling_bert.py:348-352, BertIntermediate :394-397, BertOutput :407-411)
 * and its autograd backward. */
typedef struct {
  int B, L, H, heads, I;
  float eps;
  int training;        /* save activations for backward                       */
  uint32_t p_hidden16; /* dropout threshold (p*65536) for dense outputs        */
  uint32_t p_attn16;   /* dropout threshold for attention probabilities        */
  uint64_t seed;       /* per-layer-call seed                                  */
  /* row-packed mode (see mvptr_attention_fwd_packed): M > 0 rows in total instead of B*L, sequence b
   * at rows [seq_start[b], +seq_len[b]); L = maximum length; mask_add may then be NULL.
   * M == 0 / NULL arrays: dense [B, L] layout. */
  int M;
  /* Device-side row count (ABI 4, the sync-free joint + hard-negative pass whose row count depends on the mined negatives):
   * rows_dev != NULL: device int32 holding the rows actually present (<= M).  M is then the BOUND the buffers (x, y, saved,
   * ws) are sized for; every kernel clamps to *rows_dev (workgroups past it return at once), so the host never reads the
   * count.  M_plan (0: M): rows the launches are planned for (tile configuration, M-splits) — e.g. the previous step's
   * count.  Rows [*rows_dev, M) of y / dx / the stash are left unwritten. */
  int M_plan;
  const int* seq_start;
  const int* seq_len;
  const int* rows_dev;
  /* ABI 5: 0 = gelu'(u) stashed as 8-bit fixed point (1 B per element, dithered rounding, |error| < 0.005 with zero mean: the default), 1 = as bf16 (2 B per
   * element: the stash of rounds 1-3) — must be the same in the forward and the backward call of a layer */
  int stash_bf16;
  /* ABI 5: 1 = GEMMs of another stack run beside this layer on a second stream (the text and visual stacks of the two-stage
   * model): its GEMM tiles keep their full 256-row height — under-filled rounds of the 256 CUs are filled by the other stream.
   * 0 = the launch has the GPU to itself (joint stack, one-stream jobs): tile height 256 / 224 / 192 / 160 rows by whole CU rounds. */
  int beside;
} mvptr_layer_desc;

typedef struct {
  const void* w_qkv;   /* bf16 [3H, H]  (query|key|value rows)                 */
  const void* w_qkv_t; /* bf16 [H, 3H]  transposed copy for dgrad              */
  const float* b_qkv;  /* f32 [3H]                                             */
  const void* w_o;     /* bf16 [H, H]   attention.output.dense                 */
  const void* w_o_t;   /* bf16 [H, H]                                          */
  const float* b_o;
  const float* ln1_g;  /* attention.output.LayerNorm                           */
  const float* ln1_b;
  const void* w_i;     /* bf16 [I, H]   intermediate.dense                     */
  const void* w_i_t;   /* bf16 [H, I]                                          */
  const float* b_i;
  const void* w_out;   /* bf16 [H, I]   output.dense                           */
  const void* w_out_t; /* bf16 [I, H]                                          */
  const float* b_out;
  const float* ln2_g;  /* output.LayerNorm                                     */
  const float* ln2_b;
} mvptr_layer_weights;

/* f32 gradient buffers (accumulated into; caller zeroes) */
typedef struct {
  float* w_qkv; /* [3H, H] */
  float* b_qkv;
  float* w_o;
  float* b_o;
  float* ln1_g;
  float* ln1_b;
  float* w_i;
  float* b_i;
  float* w_out;
  float* b_out;
  float* ln2_g;
  float* ln2_b;
} mvptr_layer_grads;

/* bytes of the per-layer activation stash (training) and of the scratch workspace */
int64_t mvptr_layer_saved_bytes(const mvptr_layer_desc* d);
int64_t mvptr_layer_workspace_bytes(const mvptr_layer_desc* d);

/* x: bf16 [B*L, H] -> y: bf16 [B*L, H].  saved: activation stash of mvptr_layer_saved_bytes()
 * bytes (kept for backward when training, reusable scratch otherwise); ws is unused by forward. */
int mvptr_encoder_layer_fwd(const mvptr_layer_desc* d, const mvptr_layer_weights* w,
                            const void* x, const float* mask_add, void* y, void* saved,
                            void* ws, int64_t ws_bytes, void* stream);

/* dy: bf16 [B*L,H] -> dx: bf16 [B*L,H]; weight grads accumulated into g.
 * x is the layer input given to forward (the previous layer's y). */
int mvptr_encoder_layer_bwd(const mvptr_layer_desc* d, const mvptr_layer_weights* w,
                            const void* x, const float* mask_add, const void* saved,
                            const void* dy, void* dx, const mvptr_layer_grads* g, void* ws,
                            int64_t ws_bytes, void* stream);
/* The same without the weight-gradient launches (ABI 5): the layer's (up to four) weight-gradient problems are written to
 * wgrads[0 .. *n_wgrads) for a later mvptr_gemm_tn_stack over all layers of the stack.  Their dY operands live in ws, so
 * every layer needs a workspace of its OWN that stays untouched until that launch has run (the X operands are x and the
 * activation stash); the bias gradients of intermediate.dense and of Q/K/V ride on the deferred problems (colsum). */
int mvptr_encoder_layer_bwd_defer(const mvptr_layer_desc* d, const mvptr_layer_weights* w,
                                  const void* x, const float* mask_add, const void* saved,
                                  const void* dy, void* dx, const mvptr_layer_grads* g, void* ws,
                                  int64_t ws_bytes, mvptr_tn_problem* wgrads, int* n_wgrads, void* stream);

/* The measurement helpers (mvptr_diag_*: FETCH_SIZE calibration, store / fill probes) are declared in mvptr_diag.h and exist in the
 * diagnostic build of the library only (libmvptr_hip_diag.so, `make diag`); the product library does not export them. */

#ifdef __cplusplus
}
#endif
#endif /* MVPTR_H */
