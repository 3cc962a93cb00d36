// heads.hip — the B-row tails of the cross-modal model in f32: pooler (modeling_bert.py:468-474), the 2-way
// image-text-matching head (oscar/modeling/modeling_vlbert.py:975-979,1247-1251), the CLIP-style global
// projections, L2 normalisation, similarity matrix and symmetric contrastive loss (:525-527,1238-1241).
//
// These operate on B or 2B rows ([CLS] states) — 0.1-0.6 GFLOP each, latency-bound — and stay in f32 like
// the reference (sim_mat feeds the hard-negative argmax :531-534, so it must not be rounded to bf16): exact
// f32 FMA chains on the vector ALUs, 32 x 32 output tiles through LDS.  What this file buys over library
// calls is launch count and independence from the host application's BLAS selection, not FLOP/s.
#include "common.h"

namespace {

struct SgemmArgs {
  const void* A;          // op(A) is [M, K]
  const void* B;          // op(B) is [K, N]
  float* C;               // [M, N]
  const float* bias;      // [N] or NULL
  const int32_t* a_rows;  // optional gather: row m of op(A) (or column when trans_a) is A row a_rows[m]
  const int32_t* b_rows;  // optional gather of B's stored rows (k when !trans_b, n when trans_b)
  int64_t lda, ldb, ldc;
  int M, N, K;
  int trans_a, trans_b;   // 0: as stored [M,K] / [K,N]; 1: stored transposed ([K,M] / [N,K])
  int a_bf16, b_bf16;     // element type of A / B (0: f32, 1: bf16)
  int act;                // 0: none, 1: tanh
  int accumulate;         // C += result (gradients into an arena) instead of C =
  float alpha;
};

__device__ __forceinline__ float ld_elem(const void* base, int64_t idx, int is_bf16) {
  return is_bf16 ? bf2f(((const __bf16*)base)[idx]) : ((const float*)base)[idx];
}

// C[M,N] = act(alpha * op(A) op(B) + bias): one 32 x 32 output tile per 256-thread workgroup.  The four waves
// each take a quarter of K and multiply with v_mfma_f32_32x32x2_f32 (f32 in, f32 accumulate: an fmaf chain,
// exact f32 at the vector rate) on operands loaded from global memory STRAIGHT into the MFMA operand layout — no
// LDS staging and no barrier in the loop, eight 8-deep k-chunks (32 independent loads per lane) in flight: these
// products are latency-bound (192 workgroups for a 256 x 768 output, one per CU), so what counts is how many
// loads a wave keeps outstanding.  MFMA step j of a chunk uses k = chunk + 4 h + j on lane half h for BOTH
// operands (a sum over k may take its terms in any order), so a lane's four k values are contiguous: one 16-byte
// (f32) or 8-byte (bf16) load where k is the operand's contiguous index, four row-coalesced scalar loads
// where it is not.  The partial tiles of waves 1-3 are summed through LDS at the end.
template <bool BF16>
__device__ __forceinline__ void sg_load4(const void* base, int64_t row_off, int64_t k_stride, int kk, int klimit, bool row_ok,
                                         float out[4]) {
  // four values at element offsets row_off + (kk + j) * k_stride, j = 0..3; zeros outside the problem
  if (k_stride == 1 && row_ok && kk + 3 < klimit && ((row_off + kk) & 3) == 0 && (((uintptr_t)base) & (BF16 ? 7 : 15)) == 0) {
    if (BF16) {
      const bf16x4 v = *reinterpret_cast<const bf16x4*>((const __bf16*)base + row_off + kk);
#pragma unroll
      for (int j = 0; j < 4; ++j) out[j] = bf2f(v[j]);
    } else {
      const f32x4 v = *reinterpret_cast<const f32x4*>((const float*)base + row_off + kk);
#pragma unroll
      for (int j = 0; j < 4; ++j) out[j] = v[j];
    }
    return;
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    float v = 0.f;
    if (row_ok && kk + j < klimit) {
      const int64_t o = row_off + (int64_t)(kk + j) * k_stride;
      v = BF16 ? bf2f(((const __bf16*)base)[o]) : ((const float*)base)[o];
    }
    out[j] = v;
  }
}

template <bool ABF, bool BBF>
__global__ __launch_bounds__(256) void sgemm_small_kernel(SgemmArgs p) {
  __shared__ float red[3][16][64];      // partial tiles of waves 1..3
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int m0 = blockIdx.y * 32, n0 = blockIdx.x * 32;
  const int l31 = lane & 31, hh = lane >> 5;
  const int m = m0 + l31, n = n0 + l31;
  // this wave's k range: a quarter of K rounded up to whole 8-deep chunks
  const int kq = ((p.K + 31) / 32) * 8;
  const int kbeg = wave * kq, kend = min(p.K, kbeg + kq);
  // A: lane (row m, half h) reads op(A)[m][k]; B: lane (column n, half h) reads op(B)[k][n]
  const bool a_gather_k = p.trans_a && p.a_rows != nullptr, b_gather_k = !p.trans_b && p.b_rows != nullptr;
  int64_t a_off = 0, a_ks = 1, b_off = 0, b_ks = 1;
  const bool a_ok = m < p.M, b_ok = n < p.N;
  if (p.trans_a) { a_off = m; a_ks = p.lda; }                 // stored [K, M]
  else { a_off = (a_ok ? (p.a_rows ? (int64_t)p.a_rows[m] : (int64_t)m) : 0) * p.lda; a_ks = 1; }
  if (p.trans_b) { b_off = (b_ok ? (p.b_rows ? (int64_t)p.b_rows[n] : (int64_t)n) : 0) * p.ldb; b_ks = 1; }   // stored [N, K]
  else { b_off = n; b_ks = p.ldb; }
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  constexpr int UN = 8;   // chunks in flight
  for (int kb = kbeg; kb < kend; kb += 8 * UN) {
    float av[UN][4], bv[UN][4];
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      const int kk = kb + 8 * u + 4 * hh;
      if (a_gather_k || b_gather_k) {
        // the gather indexes stored rows = k: resolve each k on its own
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int k = kk + j;
          float x = 0.f, y = 0.f;
          if (k < kend) {
            if (a_ok) {
              const int64_t o = p.trans_a ? (a_gather_k ? (int64_t)p.a_rows[k] : (int64_t)k) * p.lda + m : a_off + k;
              x = ABF ? bf2f(((const __bf16*)p.A)[o]) : ((const float*)p.A)[o];
            }
            if (b_ok) {
              const int64_t o = p.trans_b ? b_off + k : (b_gather_k ? (int64_t)p.b_rows[k] : (int64_t)k) * p.ldb + n;
              y = BBF ? bf2f(((const __bf16*)p.B)[o]) : ((const float*)p.B)[o];
            }
          }
          av[u][j] = x;
          bv[u][j] = y;
        }
      } else {
        sg_load4<ABF>(p.A, a_off, a_ks, kk, kend, a_ok, av[u]);
        sg_load4<BBF>(p.B, b_off, b_ks, kk, kend, b_ok, bv[u]);
      }
    }
#pragma unroll
    for (int u = 0; u < UN; ++u)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u][j], bv[u][j], acc, 0, 0, 0);
  }
  if (wave > 0) {
#pragma unroll
    for (int r = 0; r < 16; ++r) red[wave - 1][r][lane] = acc[r];
  }
  __syncthreads();
  if (wave != 0 || n >= p.N) return;
  const float bias = p.bias ? p.bias[n] : 0.f;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int mr = m0 + (r & 3) + 8 * (r >> 2) + 4 * hh;   // 32x32 accumulator layout: register r, lane half hh -> row
    if (mr >= p.M) continue;
    float v = ((acc[r] + red[0][r][lane]) + (red[1][r][lane] + red[2][r][lane])) * p.alpha + bias;
    if (p.act == 1) v = tanhf(v);
    float* c = p.C + (int64_t)mr * p.ldc + n;
    *c = p.accumulate ? (*c + v) : v;
  }
}

// g[r,:] = y[r,:] / max(||y[r,:]||, eps); inv[r] = 1 / max(||y||, eps)       (F.normalize(p=2, dim=-1))
__global__ __launch_bounds__(256) void l2norm_fwd_kernel(const float* y, float* g, float* inv, int H, float eps) {
  const int r = blockIdx.x;
  const float* yr = y + (int64_t)r * H;
  float s = 0.f;
  for (int c = threadIdx.x; c < H; c += 256) s += yr[c] * yr[c];
  s = wave_sum(s);
  __shared__ float red[4];
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  const float nrm = sqrtf((red[0] + red[1]) + (red[2] + red[3]));
  const float iv = 1.f / fmaxf(nrm, eps);
  for (int c = threadIdx.x; c < H; c += 256) g[(int64_t)r * H + c] = yr[c] * iv;
  if (threadIdx.x == 0) inv[r] = iv;
}

// dy = (dg - g (g . dg)) * inv   (rows with ||y|| < eps, where normalize is a plain scale, do not occur for
// projected [CLS] states; they would get the scale-only gradient dg * inv from the same formula's first term)
__global__ __launch_bounds__(256) void l2norm_bwd_kernel(const float* g, const float* inv, const float* dg, float* dy, int H) {
  const int r = blockIdx.x;
  const float* gr = g + (int64_t)r * H;
  const float* dr = dg + (int64_t)r * H;
  float s = 0.f;
  for (int c = threadIdx.x; c < H; c += 256) s += gr[c] * dr[c];
  s = wave_sum(s);
  __shared__ float red[4];
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  const float dot = (red[0] + red[1]) + (red[2] + red[3]);
  const float iv = inv[r];
  for (int c = threadIdx.x; c < H; c += 256) dy[(int64_t)r * H + c] = (dr[c] - gr[c] * dot) * iv;
}

// Symmetric contrastive loss over sim [n, n] (row i = text i, column j = image j), logits = sim * scale:
//   loss = ( mean_i CE(logits[i, :], i) + mean_j CE(logits[:, j], j) ) / 2          (vl:1238-1241)
// one workgroup per row (blocks 0..n-1) and per column (blocks n..2n-1): log-sum-exp -> lse[2n];
// parts[b] = lse - logits[i, i] (summed by clip_ce_sum_kernel in index order).
__global__ __launch_bounds__(256) void clip_ce_lse_kernel(const float* sim, int n, int64_t ld, const float* logit_scale,
                                                           float* lse, float* parts) {
  const int b = blockIdx.x;
  const bool col = b >= n;
  const int i = col ? b - n : b;
  const float sc = __expf(logit_scale[0]);
  float m = -1e30f, s = 0.f;
  for (int j = threadIdx.x; j < n; j += 256) {
    const float v = (col ? sim[(int64_t)j * ld + i] : sim[(int64_t)i * ld + j]) * sc;
    const float mm = fmaxf(m, v);
    s = s * __expf(m - mm) + __expf(v - mm);
    m = mm;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float m2 = __shfl_xor(m, o), s2 = __shfl_xor(s, o);
    const float mm = fmaxf(m, m2);
    s = s * __expf(m - mm) + s2 * __expf(m2 - mm);
    m = mm;
  }
  __shared__ float sm[8];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (lane == 0) {
    sm[wave] = m;
    sm[4 + wave] = s;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    const float mm = fmaxf(fmaxf(sm[0], sm[1]), fmaxf(sm[2], sm[3]));
    float ss = 0.f;
#pragma unroll
    for (int w = 0; w < 4; ++w) ss += sm[4 + w] * __expf(sm[w] - mm);
    const float l = mm + logf(ss);
    lse[b] = l;
    parts[b] = l - sim[(int64_t)i * ld + i] * sc;
  }
}
__global__ __launch_bounds__(256) void clip_ce_sum_kernel(const float* parts, int n, float* loss) {
  __shared__ float red[256];
  float s = 0.f;
  for (int i = threadIdx.x; i < 2 * n; i += 256) s += parts[i];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) loss[0] = red[0] / (2.f * (float)n);
}
// dlogits[i,j] = gloss / (2n) * (exp(l_ij - lse_row[i]) + exp(l_ij - lse_col[j]) - 2 [i == j]);
// dsim = dlogits * scale; d(logit_scale) += sum dlogits * logits   (logits = sim * exp(logit_scale))
__global__ __launch_bounds__(256) void clip_ce_bwd_kernel(const float* sim, int n, int64_t ld, const float* logit_scale,
                                                           const float* lse, const float* gloss, float* dsim,
                                                           float* dscale_parts) {
  const int i = blockIdx.x;
  const float sc = __expf(logit_scale[0]);
  const float w = gloss[0] / (2.f * (float)n);
  float acc = 0.f;
  for (int j = threadIdx.x; j < n; j += 256) {
    const float l = sim[(int64_t)i * ld + j] * sc;
    const float d = w * (__expf(l - lse[i]) + __expf(l - lse[n + j]) - (i == j ? 2.f : 0.f));
    dsim[(int64_t)i * n + j] = d * sc;
    acc += d * l;
  }
  acc = wave_sum(acc);
  __shared__ float red[4];
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) dscale_parts[i] = (red[0] + red[1]) + (red[2] + red[3]);
}
__global__ __launch_bounds__(256) void sum_add_kernel(const float* parts, int n, float* out) {
  __shared__ float red[256];
  float s = 0.f;
  for (int i = threadIdx.x; i < n; i += 256) s += parts[i];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[0] += red[0];
}

}  // namespace

extern "C" int mvptr_sgemm_small(const void* A, int64_t lda, int a_bf16, int trans_a, const int32_t* a_rows, const void* B,
                                 int64_t ldb, int b_bf16, int trans_b, const int32_t* b_rows, int M, int N, int K, float alpha, const float* bias,
                                 int act, int accumulate, float* C, int64_t ldc, void* stream) {
  if (M <= 0 || N <= 0 || K <= 0) MVPTR_FAIL(MVPTR_BAD_SHAPE, "sgemm_small: M, N, K must be > 0");
  if (!A || !B || !C) MVPTR_FAIL(MVPTR_BAD_ARG, "sgemm_small: NULL argument");
  if (act < 0 || act > 1) MVPTR_FAIL(MVPTR_BAD_ARG, "sgemm_small: unknown activation %d", act);
  SgemmArgs p;
  p.A = A; p.B = B; p.C = C; p.bias = bias; p.a_rows = a_rows; p.b_rows = b_rows;
  p.lda = lda; p.ldb = ldb; p.ldc = ldc;
  p.M = M; p.N = N; p.K = K;
  p.trans_a = trans_a ? 1 : 0; p.trans_b = trans_b ? 1 : 0;
  p.a_bf16 = a_bf16 ? 1 : 0; p.b_bf16 = b_bf16 ? 1 : 0;
  p.act = act; p.accumulate = accumulate ? 1 : 0; p.alpha = alpha;
  const dim3 grid((N + 31) / 32, (M + 31) / 32);
  if (p.a_bf16 && p.b_bf16) hipLaunchKernelGGL((sgemm_small_kernel<true, true>), grid, dim3(256), 0, (hipStream_t)stream, p);
  else if (p.a_bf16) hipLaunchKernelGGL((sgemm_small_kernel<true, false>), grid, dim3(256), 0, (hipStream_t)stream, p);
  else if (p.b_bf16) hipLaunchKernelGGL((sgemm_small_kernel<false, true>), grid, dim3(256), 0, (hipStream_t)stream, p);
  else hipLaunchKernelGGL((sgemm_small_kernel<false, false>), grid, dim3(256), 0, (hipStream_t)stream, p);
  MVPTR_CHECK_LAUNCH("sgemm_small");
  return MVPTR_OK;
}

extern "C" int mvptr_l2norm_fwd(const float* y, float* g, float* inv_norm, int rows, int H, float eps, void* stream) {
  if (rows <= 0 || H <= 0 || !y || !g || !inv_norm) MVPTR_FAIL(MVPTR_BAD_ARG, "l2norm_fwd: bad argument");
  hipLaunchKernelGGL(l2norm_fwd_kernel, dim3(rows), dim3(256), 0, (hipStream_t)stream, y, g, inv_norm, H, eps);
  MVPTR_CHECK_LAUNCH("l2norm_fwd");
  return MVPTR_OK;
}

extern "C" int mvptr_l2norm_bwd(const float* g, const float* inv_norm, const float* dg, float* dy, int rows, int H, void* stream) {
  if (rows <= 0 || H <= 0 || !g || !inv_norm || !dg || !dy) MVPTR_FAIL(MVPTR_BAD_ARG, "l2norm_bwd: bad argument");
  hipLaunchKernelGGL(l2norm_bwd_kernel, dim3(rows), dim3(256), 0, (hipStream_t)stream, g, inv_norm, dg, dy, H);
  MVPTR_CHECK_LAUNCH("l2norm_bwd");
  return MVPTR_OK;
}

extern "C" int mvptr_clip_ce_fwd(const float* sim, int n, int64_t ld, const float* logit_scale, float* lse, float* parts,
                                 float* loss, void* stream) {
  if (n <= 0 || ld < n || !sim || !logit_scale || !lse || !parts || !loss) MVPTR_FAIL(MVPTR_BAD_ARG, "clip_ce_fwd: bad argument");
  hipLaunchKernelGGL(clip_ce_lse_kernel, dim3(2 * n), dim3(256), 0, (hipStream_t)stream, sim, n, ld, logit_scale, lse, parts);
  hipLaunchKernelGGL(clip_ce_sum_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, parts, n, loss);
  MVPTR_CHECK_LAUNCH("clip_ce_fwd");
  return MVPTR_OK;
}

extern "C" int mvptr_clip_ce_bwd(const float* sim, int n, int64_t ld, const float* logit_scale, const float* lse,
                                 const float* gloss, float* dsim, float* parts, float* dlogit_scale, void* stream) {
  if (n <= 0 || ld < n || !sim || !logit_scale || !lse || !gloss || !dsim || !parts)
    MVPTR_FAIL(MVPTR_BAD_ARG, "clip_ce_bwd: bad argument");
  hipLaunchKernelGGL(clip_ce_bwd_kernel, dim3(n), dim3(256), 0, (hipStream_t)stream, sim, n, ld, logit_scale, lse, gloss, dsim, parts);
  if (dlogit_scale != nullptr)
    hipLaunchKernelGGL(sum_add_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, parts, n, dlogit_scale);
  MVPTR_CHECK_LAUNCH("clip_ce_bwd");
  return MVPTR_OK;
}

namespace {
// mean cross entropy over M rows of V <= 64 classes (ITM: V = 2), rows with label < 0 or >= V ignored
// (CrossEntropyLoss(ignore_index=-1), vl:1247-1251): loss[0] = sum / max(count, 1);
// dlogits[m, v] = (softmax - onehot) / max(count, 1), 0 for ignored rows.  One workgroup.
__global__ __launch_bounds__(256) void ce_mean_small_kernel(const float* logits, int64_t ld, const int64_t* labels, int M, int V,
                                                             float* loss, float* dlogits) {
  __shared__ float red[256];
  __shared__ int cnt[256];
  float s = 0.f;
  int c = 0;
  for (int m = threadIdx.x; m < M; m += 256) {
    const int64_t lab = labels[m];
    if (lab < 0 || lab >= V) continue;
    const float* x = logits + (int64_t)m * ld;
    float mx = x[0];
    for (int v = 1; v < V; ++v) mx = fmaxf(mx, x[v]);
    float se = 0.f;
    for (int v = 0; v < V; ++v) se += __expf(x[v] - mx);
    s += mx + logf(se) - x[lab];
    ++c;
  }
  red[threadIdx.x] = s;
  cnt[threadIdx.x] = c;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) {
      red[threadIdx.x] += red[threadIdx.x + o];
      cnt[threadIdx.x] += cnt[threadIdx.x + o];
    }
    __syncthreads();
  }
  const float inv = 1.f / (float)max(cnt[0], 1);
  if (threadIdx.x == 0) loss[0] = red[0] * inv;
  if (dlogits == nullptr) return;
  for (int m = threadIdx.x; m < M; m += 256) {
    const int64_t lab = labels[m];
    const bool ok = lab >= 0 && lab < V;
    const float* x = logits + (int64_t)m * ld;
    float mx = x[0];
    for (int v = 1; v < V; ++v) mx = fmaxf(mx, x[v]);
    float se = 0.f;
    for (int v = 0; v < V; ++v) se += __expf(x[v] - mx);
    const float r = 1.f / se;
    for (int v = 0; v < V; ++v)
      dlogits[(int64_t)m * V + v] = ok ? (__expf(x[v] - mx) * r - ((int64_t)v == lab ? 1.f : 0.f)) * inv : 0.f;
  }
}
}  // namespace

extern "C" int mvptr_ce_mean_small(const float* logits, int64_t ld, const int64_t* labels, int M, int V, float* loss,
                                   float* dlogits, void* stream) {
  if (M <= 0 || V <= 0 || V > 64 || ld < V || !logits || !labels || !loss) MVPTR_FAIL(MVPTR_BAD_ARG, "ce_mean_small: bad argument (V <= 64)");
  hipLaunchKernelGGL(ce_mean_small_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, logits, ld, labels, M, V, loss, dlogits);
  MVPTR_CHECK_LAUNCH("ce_mean_small");
  return MVPTR_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// Two pieces of head glue that were chains of tiny torch kernels (round 4: 6 + 8 launches per MLM head and step):
//  * masked mean of the per-row losses of the fused decoder + cross-entropy:  out[0] = sum(loss_row) / max(count, 1),
//    out[1] = max(count, 1), count = rows with label >= 0 (CrossEntropyLoss(ignore_index=-1) of vl:1247-1251: rows the
//    kernels did not score carry loss_row = 0).  One workgroup, fixed tree: bitwise reproducible.
//  * d(gelu) -> du of a head transform: out = bf16(dy * gelu'(u)), gelu'(u) decoded from the 8-bit stash the forward
//    epilogue wrote (common.h); columns N .. Npad of out (operand padding of the next GEMM) are zeroed.
namespace {
__global__ __launch_bounds__(1024) void masked_mean_kernel(const float* loss_row, const int64_t* labels, int M, float* out) {
  __shared__ float red[1024];
  __shared__ int cnt[1024];
  float s = 0.f;
  int c = 0;
  for (int m = threadIdx.x; m < M; m += 1024) {
    s += loss_row[m];
    c += labels[m] >= 0 ? 1 : 0;
  }
  red[threadIdx.x] = s;
  cnt[threadIdx.x] = c;
  __syncthreads();
  for (int o = 512; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) {
      red[threadIdx.x] += red[threadIdx.x + o];
      cnt[threadIdx.x] += cnt[threadIdx.x + o];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const float n = (float)max(cnt[0], 1);
    out[0] = red[0] / n;
    out[1] = n;
  }
}
template <bool BF16>
__global__ __launch_bounds__(256) void dgelu_mul_kernel(const __bf16* dy, int64_t ld_dy, const uint8_t* stash, int64_t ld_s, __bf16* out,
                                                         int64_t ld_o, int M, int N, int Npad) {
  const int quads = Npad >> 2;
  const int64_t total = (int64_t)M * quads;
  for (int64_t e = blockIdx.x * (int64_t)256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
    const int m = (int)(e / quads), n = (int)(e - (int64_t)m * quads) * 4;
    bf16x4 o = {f2bf(0.f), f2bf(0.f), f2bf(0.f), f2bf(0.f)};
#pragma unroll
    for (int k = 0; k < 4; ++k)
      if (n + k < N) {
        const float g = BF16 ? bf2f(reinterpret_cast<const __bf16*>(stash)[(int64_t)m * ld_s + n + k]) : dgelu_decode1(stash[(int64_t)m * ld_s + n + k]);
        o[k] = f2bf(bf2f(dy[(int64_t)m * ld_dy + n + k]) * g);
      }
    *reinterpret_cast<bf16x4*>(out + (int64_t)m * ld_o + n) = o;
  }
}
}  // namespace

extern "C" int mvptr_masked_mean(const float* loss_row, const int64_t* labels, int M, float* out, void* stream) {
  if (M <= 0 || !loss_row || !labels || !out) MVPTR_FAIL(MVPTR_BAD_ARG, "masked_mean: bad argument");
  hipLaunchKernelGGL(masked_mean_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, loss_row, labels, M, out);
  MVPTR_CHECK_LAUNCH("masked_mean");
  return MVPTR_OK;
}

extern "C" int mvptr_dgelu_mul(const void* dy, int64_t ld_dy, const void* stash, int64_t ld_s, int stash_bf16, void* out, int64_t ld_o, int M,
                               int N, int Npad, void* stream) {
  if (M <= 0 || N <= 0 || Npad < N || (Npad & 3) || ld_dy < N || ld_s < N || ld_o < Npad || (ld_o & 3) || !dy || !stash || !out ||
      ((uintptr_t)out & 7))
    MVPTR_FAIL(MVPTR_BAD_ARG, "dgelu_mul: bad shape / alignment (Npad %% 4 == 0, ld_o >= Npad, out 8-byte aligned)");
  const int64_t total = (int64_t)M * (Npad >> 2);
  int grid = (int)((total + 255) / 256);
  if (grid > 8192) grid = 8192;
  if (stash_bf16)
    hipLaunchKernelGGL(dgelu_mul_kernel<true>, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const __bf16*)dy, ld_dy, (const uint8_t*)stash,
                       ld_s, (__bf16*)out, ld_o, M, N, Npad);
  else
    hipLaunchKernelGGL(dgelu_mul_kernel<false>, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const __bf16*)dy, ld_dy, (const uint8_t*)stash,
                       ld_s, (__bf16*)out, ld_o, M, N, Npad);
  MVPTR_CHECK_LAUNCH("dgelu_mul");
  return MVPTR_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// In-batch hard negatives (oscar/modeling/modeling_vlbert.py:529-566, hn_mod = 'hard'): per text the most similar
// OTHER image, per image the most similar other text (argmax of sim - 2 I along rows / columns), then the two halves of
// a random permutation decide which side of each hard pair is replaced:
//   hard_txt_full = [dice[:n/2] ; hard_txt[dice[n/2:]]],  hard_img_full = [hard_img[dice[:n/2]] ; dice[n/2:]]
// (the `dice` = torch.randperm(n) stays torch's draw, so torch.manual_seed controls it and the parity tests replay it).
// Also writes the `sel` vectors of the joint + hard-negative pack maps, [0..n) ++ hard_*_full.  Two launches instead of
// the ~25 small torch kernels (eye, sub, 2 x max, 4 x index_select, arange, 4 x cat ...).  Ties: the lowest index wins.
namespace {
__global__ __launch_bounds__(256) void hn_argmax_kernel(const float* sim, int n, int64_t ld, int64_t* hard_img, int64_t* hard_txt) {
  const int w = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (w >= 2 * n) return;
  const bool col = w >= n;
  const int i = col ? w - n : w;
  float best = -INFINITY;
  int arg = n;
  for (int j = lane; j < n; j += 64) {
    float v = col ? sim[(int64_t)j * ld + i] : sim[(int64_t)i * ld + j];
    if (j == i) v -= 2.0f;
    if (v > best) best = v, arg = j;       // ascending j per lane: the first maximum stays
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float b2 = __shfl_xor(best, o);
    const int a2 = __shfl_xor(arg, o);
    if (b2 > best || (b2 == best && a2 < arg)) best = b2, arg = a2;
  }
  if (lane == 0) (col ? hard_txt : hard_img)[i] = arg < n ? arg : 0;
}

__global__ __launch_bounds__(256) void hn_select_kernel(const int64_t* perm, const int64_t* hard_img, const int64_t* hard_txt, int n,
                                                        int64_t* hard_txt_full, int64_t* hard_img_full, int64_t* sel_txt, int64_t* sel_img) {
  const int k = blockIdx.x * 256 + threadIdx.x;
  if (k >= n) return;
  int64_t d = perm[k];
  d = d < 0 ? 0 : (d >= n ? n - 1 : d);
  const bool first = k < n / 2;
  const int64_t t = first ? d : hard_txt[d], i = first ? hard_img[d] : d;
  hard_txt_full[k] = t;
  hard_img_full[k] = i;
  if (sel_txt) sel_txt[k] = k, sel_txt[n + k] = t;
  if (sel_img) sel_img[k] = k, sel_img[n + k] = i;
}

// sum over all elements of max(x, 0) - x y + log(1 + exp(-|x|)), divided by `rows` = instance_bce_with_logits
// (vl:878-883: binary_cross_entropy_with_logits(mean) * labels.size(1)); dlogits = (sigmoid(x) - y) / rows.
__global__ __launch_bounds__(256) void bce_partial_kernel(const float* x, const float* y, int64_t total, float inv_rows, float* dx, float* part) {
  __shared__ float red[256];
  float s = 0.f;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
    const float v = x[e], t = y[e];
    const float ex = __expf(-fabsf(v));
    s += fmaxf(v, 0.f) - v * t + log1pf(ex);
    if (dx) {
      const float sig = v >= 0.f ? 1.f / (1.f + ex) : ex / (1.f + ex);
      dx[e] = (sig - t) * inv_rows;
    }
  }
  red[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) part[blockIdx.x] = red[0];
}
__global__ __launch_bounds__(256) void bce_final_kernel(const float* part, int n, float inv_rows, float* loss) {
  __shared__ float red[256];
  float s = 0.f;
  for (int i = threadIdx.x; i < n; i += 256) s += part[i];     // fixed order: reproducible
  red[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) loss[0] = red[0] * inv_rows;
}
}  // namespace

extern "C" int mvptr_hard_negative_mine(const float* sim, int n, int64_t ld, const int64_t* perm, int64_t* hard_img, int64_t* hard_txt,
                                        int64_t* hard_txt_full, int64_t* hard_img_full, int64_t* sel_txt, int64_t* sel_img, void* stream) {
  if (n <= 0 || ld < n || !sim || !hard_img || !hard_txt) MVPTR_FAIL(MVPTR_BAD_ARG, "hard_negative_mine: bad argument");
  if (perm != nullptr && (!hard_txt_full || !hard_img_full)) MVPTR_FAIL(MVPTR_BAD_ARG, "hard_negative_mine: perm without outputs");
  hipLaunchKernelGGL(hn_argmax_kernel, dim3((2 * n + 3) / 4), dim3(256), 0, (hipStream_t)stream, sim, n, ld, hard_img, hard_txt);
  if (perm != nullptr)
    hipLaunchKernelGGL(hn_select_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, perm, hard_img, hard_txt, n,
                       hard_txt_full, hard_img_full, sel_txt, sel_img);
  MVPTR_CHECK_LAUNCH("hard_negative_mine");
  return MVPTR_OK;
}

extern "C" int mvptr_bce_logits(const float* logits, const float* labels, int rows, int cols, float* loss, float* dlogits,
                                float* parts, int n_parts, void* stream) {
  if (rows <= 0 || cols <= 0 || !logits || !labels || !loss || !parts || n_parts < 1 || n_parts > 4096)
    MVPTR_FAIL(MVPTR_BAD_ARG, "bce_logits: bad argument (1 <= n_parts <= 4096)");
  const float inv = 1.0f / (float)rows;
  hipLaunchKernelGGL(bce_partial_kernel, dim3(n_parts), dim3(256), 0, (hipStream_t)stream, logits, labels, (int64_t)rows * cols, inv, dlogits, parts);
  hipLaunchKernelGGL(bce_final_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, parts, n_parts, inv, loss);
  MVPTR_CHECK_LAUNCH("bce_logits");
  return MVPTR_OK;
}
