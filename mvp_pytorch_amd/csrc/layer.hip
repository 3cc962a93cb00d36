// layer.hip — one BERT encoder layer (forward / backward) composed from the kernels of this
// library on a single HIP stream.
//
// Replaces CaptionBertLayer.forward oscar/modeling/modeling_vlbert.py:191-199 =
//   CaptionBertSelfAttention :63-103 -> BertSelfOutput modeling_bert.py:348-352 ->
//   BertIntermediate :394-397 -> BertOutput :407-411, and the autograd backward of that chain.
#include "common.h"

namespace {

inline int64_t al256(int64_t x) { return (x + 255) & ~(int64_t)255; }

struct Stash {
  char *qkv, *ctx, *z1, *x1, *u, *a, *z2;
  float *lse, *mean1, *rstd1, *mean2, *rstd2;
  int64_t total;
};

Stash carve(const mvptr_layer_desc* d, void* base) {
  const int64_t M = (d->M > 0 ? (int64_t)d->M : (int64_t)d->B * d->L), H = d->H, I = d->I;
  char* p = (char*)base;
  int64_t off = 0;
  Stash s;
  auto take = [&](int64_t bytes) {
    char* r = p ? p + off : nullptr;
    off += al256(bytes);
    return r;
  };
  s.qkv = take(M * 3 * H * 2);
  s.ctx = take(M * H * 2);
  s.z1 = take(M * H * 2);
  s.x1 = take(M * H * 2);
  s.u = take(M * I * (d->stash_bf16 ? 2 : 1));   // gelu'(u): 8-bit fixed point (common.h, dgelu_pack4) or bf16 (desc.stash_bf16)
  s.a = take(M * I * 2);
  s.z2 = take(M * H * 2);
  s.lse = (float*)take((int64_t)d->B * d->heads * d->L * 4);
  s.mean1 = (float*)take(M * 4);
  s.rstd1 = (float*)take(M * 4);
  s.mean2 = (float*)take(M * 4);
  s.rstd2 = (float*)take(M * 4);
  s.total = off;
  return s;
}

// token rows of a layer call: packed mode carries them in d->M, the dense layout has B * L
int64_t layer_rows(const mvptr_layer_desc* d) { return d->M > 0 ? (int64_t)d->M : (int64_t)d->B * d->L; }

mvptr_dropout site_drop(const mvptr_layer_desc* d, int site, uint32_t thresh) {
  mvptr_dropout r;
  r.seed_lo = (uint32_t)(d->seed & 0xffffffffu) ^ (0x9E3779B9u * (uint32_t)(site + 1));
  r.seed_hi = (uint32_t)(d->seed >> 32) + 0x85EBCA6Bu * (uint32_t)(site + 1);
  r.thresh16 = d->training ? thresh : 0;
  r.pad_ = 0;
  return r;
}

int check_desc(const char* who, const mvptr_layer_desc* d) {
  if (!d) MVPTR_FAIL(MVPTR_BAD_ARG, "%s: desc is NULL", who);
  if (d->B <= 0 || d->L <= 0 || d->H <= 0 || d->heads <= 0 || d->I <= 0)
    MVPTR_FAIL(MVPTR_BAD_SHAPE, "%s: non-positive dimension", who);
  if (d->H != d->heads * 64) MVPTR_FAIL(MVPTR_BAD_SHAPE, "%s: head_dim must be 64 (H=%d heads=%d)", who, d->H, d->heads);
  if ((d->H & 7) || (d->I & 7) || d->H > 1024) MVPTR_FAIL(MVPTR_BAD_SHAPE, "%s: H,I must be multiples of 8, H <= 1024", who);
  if (d->L > 256) MVPTR_FAIL(MVPTR_BAD_SHAPE, "%s: L=%d > 256", who, d->L);
  if (d->M < 0 || (int64_t)d->M > (int64_t)d->B * d->L) MVPTR_FAIL(MVPTR_BAD_SHAPE, "%s: M=%d outside [0, B*L]", who, d->M);
  if ((d->M > 0) != (d->seq_start != nullptr) || (d->M > 0) != (d->seq_len != nullptr))
    MVPTR_FAIL(MVPTR_BAD_ARG, "%s: M, seq_start and seq_len go together (row-packed mode)", who);
  if (d->rows_dev != nullptr && d->M == 0) MVPTR_FAIL(MVPTR_BAD_ARG, "%s: rows_dev needs the row-packed mode (M = the bound)", who);
  if (d->M_plan < 0 || d->M_plan > d->M) MVPTR_FAIL(MVPTR_BAD_ARG, "%s: M_plan outside [0, M]", who);
  return MVPTR_OK;
}

#define RUN(expr)            \
  do {                       \
    int rc__ = (expr);       \
    if (rc__ != 0) return rc__; \
  } while (0)

}  // namespace

extern "C" int64_t mvptr_layer_saved_bytes(const mvptr_layer_desc* d) {
  if (check_desc("layer_saved_bytes", d)) return -1;
  return carve(d, nullptr).total;
}

namespace {
// slab workspace the two grouped weight-gradient launches of a layer may use (mvptr_gemm_tn_ws_bytes: 0 when they write
// out with atomics): planned from the shapes alone
int64_t wgrad_slab_bytes(int M, int H, int I) {
  mvptr_tn_problem q[2];
  auto fill = [&](mvptr_tn_problem& t, int N, int K) {
    t.A = t.B = (const void*)(uintptr_t)4096;   // shape-only planning: the pointers are never dereferenced
    t.dW = (float*)(uintptr_t)4096;
    t.lda = N;
    t.ldb = K;
    t.M = M;
    t.N = N;
    t.K = K;
    t.ldw = K;
    t.colsum = nullptr;
  };
  fill(q[0], H, I);
  fill(q[1], I, H);
  const int64_t a = mvptr_gemm_tn_ws_bytes(q, 2);
  fill(q[0], 3 * H, H);
  fill(q[1], H, H);
  const int64_t b = mvptr_gemm_tn_ws_bytes(q, 2);
  return a > b ? a : b;
}
}  // namespace

extern "C" int64_t mvptr_layer_workspace_bytes(const mvptr_layer_desc* d) {
  if (check_desc("layer_workspace_bytes", d)) return -1;
  const int64_t M = layer_rows(d), H = d->H;
  const int64_t W = d->I > 3 * H ? d->I : 3 * H;
  // backward: five [M,H] buffers, dU [M,I], dqkv [M,3H] (every weight-gradient operand stays alive
  // until the grouped weight-gradient launch at the end of the layer) + LayerNorm partials
  (void)W;
  return 5 * al256(M * H * 2) + al256(M * d->I * 2) + al256(M * 3 * H * 2) +
         2 * al256(mvptr_layernorm_bwd_ws_bytes((int)M, (int)H)) + al256(wgrad_slab_bytes((int)M, (int)H, d->I));
}

extern "C" int mvptr_encoder_layer_fwd(const mvptr_layer_desc* d, const mvptr_layer_weights* w,
                                       const void* x, const float* mask_add, void* y, void* saved,
                                       void* ws, int64_t ws_bytes, void* stream) {
  (void)ws;
  (void)ws_bytes;
  RUN(check_desc("encoder_layer_fwd", d));
  if (!w || !x || !y || !saved || (!mask_add && d->M == 0)) MVPTR_FAIL(MVPTR_BAD_ARG, "encoder_layer_fwd: NULL argument");
  const int M = (int)layer_rows(d), H = d->H, I = d->I;
  Stash s = carve(d, saved);
  const mvptr_dropout dr_attn = site_drop(d, 0, d->p_attn16);
  const mvptr_dropout dr_o = site_drop(d, 1, d->p_hidden16);
  const mvptr_dropout dr_out = site_drop(d, 2, d->p_hidden16);
  const int* rd = d->rows_dev;     // device-side row count of a row-packed pass (NULL: M rows)
  const int Mp = d->M_plan;
  RUN(mvptr_gemm_nt_rows(x, H, w->w_qkv, H, M, 3 * H, H, MVPTR_EPI_BIAS, w->b_qkv, nullptr, 0, s.qkv,
                         nullptr, 3 * H, nullptr, nullptr, rd, Mp, stream, d->beside));
  RUN(mvptr_attention_fwd_packed(s.qkv, mask_add, s.ctx, s.lse, d->seq_start, d->seq_len, d->B, d->L, d->heads,
                                 &dr_attn, stream));
  RUN(mvptr_gemm_nt_rows(s.ctx, H, w->w_o, H, M, H, H, MVPTR_EPI_BIAS_RESID, w->b_o, x, H, s.z1, nullptr,
                         H, nullptr, &dr_o, rd, Mp, stream, d->beside));
  RUN(mvptr_layernorm_fwd_rows(s.z1, w->ln1_g, w->ln1_b, d->eps, s.x1, s.mean1, s.rstd1, M, H, M, 0, 0,
                               nullptr, rd, stream));
  RUN(mvptr_gemm_nt_rows(s.x1, H, w->w_i, H, M, I, H, d->stash_bf16 ? MVPTR_EPI_BIAS_GELU_BF16 : MVPTR_EPI_BIAS_GELU, w->b_i, nullptr, 0, s.u, s.a, I,
                         nullptr, nullptr, rd, Mp, stream, d->beside));
  RUN(mvptr_gemm_nt_rows(s.a, I, w->w_out, I, M, H, I, MVPTR_EPI_BIAS_RESID, w->b_out, s.x1, H, s.z2,
                         nullptr, H, nullptr, &dr_out, rd, Mp, stream, d->beside));
  RUN(mvptr_layernorm_fwd_rows(s.z2, w->ln2_g, w->ln2_b, d->eps, y, s.mean2, s.rstd2, M, H, M, 0, 0,
                               nullptr, rd, stream));
  return MVPTR_OK;
}

namespace {
// defer_out != NULL: no weight-gradient launches — the problems are appended to defer_out[0 .. *n_defer) instead
int layer_bwd(const mvptr_layer_desc* d, const mvptr_layer_weights* w, const void* x, const float* mask_add, const void* saved,
              const void* dy, void* dx, const mvptr_layer_grads* g, void* ws, int64_t ws_bytes, mvptr_tn_problem* defer_out,
              int* n_defer, void* stream) {
  RUN(check_desc("encoder_layer_bwd", d));
  if (!w || !x || !saved || !dy || !dx || !g || !ws || (!mask_add && d->M == 0))
    MVPTR_FAIL(MVPTR_BAD_ARG, "encoder_layer_bwd: NULL argument");
  if (ws_bytes < mvptr_layer_workspace_bytes(d))
    MVPTR_FAIL(MVPTR_WORKSPACE_TOO_SMALL, "encoder_layer_bwd: workspace %ld < %ld", (long)ws_bytes,
               (long)mvptr_layer_workspace_bytes(d));
  const int M = (int)layer_rows(d), H = d->H, I = d->I;
  Stash s = carve(d, const_cast<void*>(saved));
  char* p = (char*)ws;
  char* bufA = p;                                       // LN2 dz (residual branch)
  char* bufB = bufA + al256((int64_t)M * H * 2);        // LN2 dz, dropout-masked (dense branch)
  char* bufC = bufB + al256((int64_t)M * H * 2);        // gradient flowing down the residual stream
  char* bufD = bufC + al256((int64_t)M * H * 2);        // LN1 dz
  char* bufE = bufD + al256((int64_t)M * H * 2);        // LN1 dz, dropout-masked
  char* bufU = bufE + al256((int64_t)M * H * 2);        // dU [M, I]
  char* bufQ = bufU + al256((int64_t)M * I * 2);        // dqkv [M, 3H]
  char* lnws = bufQ + al256((int64_t)M * 3 * H * 2);
  const int64_t lnws_bytes = mvptr_layernorm_bwd_ws_bytes(M, H);
  char* lnws1 = lnws + al256(lnws_bytes);               // second LayerNorm's partials: both are finalized in one launch
  char* slabws = lnws1 + al256(lnws_bytes);             // per-split partial tiles of the weight-gradient launches (few-row stacks)
  const int64_t slabws_bytes = wgrad_slab_bytes(M, H, I);
  const mvptr_dropout dr_attn = site_drop(d, 0, d->p_attn16);
  const mvptr_dropout dr_o = site_drop(d, 1, d->p_hidden16);
  const mvptr_dropout dr_out = site_drop(d, 2, d->p_hidden16);
  const bool hdrop = dr_o.thresh16 != 0;
  // Weight gradients are collected and issued as grouped launches (FFN pair, attention pair): the
  // atomic write-out of one problem overlaps the MFMA loop of the next.
  mvptr_tn_problem wg[4];
  int nwg = 0;
  const bool defer = defer_out != nullptr;
  int ndef = 0;
  auto add_wgrad = [&](const void* dyp, int64_t lda, const void* xp, int64_t ldb, int N, int K, float* dw,
                       float* colsum) {
    mvptr_tn_problem& q = defer ? defer_out[ndef++] : wg[nwg++];
    q.A = dyp;
    q.lda = lda;
    q.B = xp;
    q.ldb = ldb;
    q.M = M;
    q.N = N;
    q.K = K;
    q.dW = dw;
    q.ldw = K;
    q.colsum = colsum;
  };

  // output.LayerNorm / output.dense
  const int* rd = d->rows_dev;     // device-side row count of a row-packed pass (NULL: M rows)
  const int Mp = d->M_plan;
  mvptr_ln_pending pend2, pend1;
  RUN(mvptr_layernorm_bwd_partial(dy, s.z2, s.mean2, s.rstd2, w->ln2_g, bufA, hdrop ? bufB : nullptr,
                                  g->ln2_g, g->ln2_b, g->b_out, M, H, M, 0, 0, nullptr,
                                  hdrop ? &dr_out : nullptr, lnws, lnws_bytes, rd, &pend2, stream));
  const char* d2 = hdrop ? bufB : bufA;
  if (g->w_out) add_wgrad(d2, H, s.a, I, H, I, g->w_out, nullptr);
  // bias gradient of intermediate.dense = column sums of dU: in the GELU-backward epilogue, or — deferred — on the weight-gradient
  // problem that reads dU anyway
  const bool bi_rides = defer && g->w_i != nullptr;
  RUN(mvptr_gemm_nt_rows(d2, H, w->w_out_t, H, M, I, H, d->stash_bf16 ? MVPTR_EPI_GELU_BWD_BF16 : MVPTR_EPI_GELU_BWD, nullptr, s.u, I, bufU, nullptr,
                         I, bi_rides ? nullptr : g->b_i, nullptr, rd, Mp, stream, d->beside));
  // intermediate.dense; the two FFN weight gradients go out together while d2 / dU are still warm
  // in the Infinity Cache
  if (g->w_i) add_wgrad(bufU, I, s.x1, H, I, H, g->w_i, bi_rides ? g->b_i : nullptr);
  if (nwg > 0) RUN(mvptr_gemm_tn_multi_rows(wg, nwg, slabws_bytes ? slabws : nullptr, slabws_bytes, rd, Mp, stream));
  nwg = 0;
  RUN(mvptr_gemm_nt_rows(bufU, I, w->w_i_t, I, M, H, I, MVPTR_EPI_ADD, nullptr, bufA, H, bufC, nullptr, H,
                         nullptr, nullptr, rd, Mp, stream, d->beside));
  // attention.output.LayerNorm / dense
  RUN(mvptr_layernorm_bwd_partial(bufC, s.z1, s.mean1, s.rstd1, w->ln1_g, bufD, hdrop ? bufE : nullptr,
                                  g->ln1_g, g->ln1_b, g->b_o, M, H, M, 0, 0, nullptr,
                                  hdrop ? &dr_o : nullptr, lnws1, lnws_bytes, rd, &pend1, stream));
  RUN(mvptr_layernorm_bwd_finalize2(&pend2, &pend1, H, stream));     // gamma / beta / bias gradients of both LayerNorms
  const char* d1 = hdrop ? bufE : bufD;
  if (g->w_o) add_wgrad(d1, H, s.ctx, H, H, H, g->w_o, nullptr);
  RUN(mvptr_gemm_nt_rows(d1, H, w->w_o_t, H, M, H, H, MVPTR_EPI_ADD, nullptr, nullptr, 0, bufC, nullptr, H,
                         nullptr, nullptr, rd, Mp, stream, d->beside));
  // attention core
  RUN(mvptr_attention_bwd_packed(s.qkv, mask_add, s.ctx, bufC, s.lse, bufQ, d->seq_start, d->seq_len, d->B,
                                 d->L, d->heads, &dr_attn, stream));
  // Q/K/V projections: the bias gradient (column sums of dqkv) rides on the weight-gradient kernel
  if (g->w_qkv) {
    add_wgrad(bufQ, 3 * H, x, H, 3 * H, H, g->w_qkv, g->b_qkv);
  } else if (g->b_qkv) {
    if (rd) MVPTR_FAIL(MVPTR_BAD_ARG, "encoder_layer_bwd: a device-side row count needs the Q/K/V weight gradient (b_qkv rides on it)");
    RUN(mvptr_colsum(bufQ, 3 * H, M, 3 * H, g->b_qkv, stream));
  }
  RUN(mvptr_gemm_nt_rows(bufQ, 3 * H, w->w_qkv_t, 3 * H, M, H, 3 * H, MVPTR_EPI_ADD, nullptr, bufD, H, dx,
                         nullptr, H, nullptr, nullptr, rd, Mp, stream, d->beside));
  // largest problem first: the exposed atomic write-out at the end of the launch is then the small one's
  if (nwg == 2 && (int64_t)wg[0].N * wg[0].K < (int64_t)wg[1].N * wg[1].K) {
    const mvptr_tn_problem t = wg[0];
    wg[0] = wg[1];
    wg[1] = t;
  }
  if (nwg > 0) RUN(mvptr_gemm_tn_multi_rows(wg, nwg, slabws_bytes ? slabws : nullptr, slabws_bytes, rd, Mp, stream));
  if (defer) *n_defer = ndef;
  return MVPTR_OK;
}
}  // namespace

extern "C" int mvptr_encoder_layer_bwd(const mvptr_layer_desc* d, const mvptr_layer_weights* w,
                                       const void* x, const float* mask_add, const void* saved,
                                       const void* dy, void* dx, const mvptr_layer_grads* g,
                                       void* ws, int64_t ws_bytes, void* stream) {
  return layer_bwd(d, w, x, mask_add, saved, dy, dx, g, ws, ws_bytes, nullptr, nullptr, stream);
}

extern "C" int mvptr_encoder_layer_bwd_defer(const mvptr_layer_desc* d, const mvptr_layer_weights* w,
                                             const void* x, const float* mask_add, const void* saved,
                                             const void* dy, void* dx, const mvptr_layer_grads* g,
                                             void* ws, int64_t ws_bytes, mvptr_tn_problem* wgrads, int* n_wgrads, void* stream) {
  if (!wgrads || !n_wgrads) MVPTR_FAIL(MVPTR_BAD_ARG, "encoder_layer_bwd_defer: wgrads / n_wgrads is NULL");
  return layer_bwd(d, w, x, mask_add, saved, dy, dx, g, ws, ws_bytes, wgrads, n_wgrads, stream);
}
