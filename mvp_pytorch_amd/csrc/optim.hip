// optim.hip — optimizer step of the pre-training loop: fused multi-tensor AdamW with the
// numerics of transformers/pytorch_transformers/optimization.py:131-187, the global gradient-norm clip of
// oscar/run_pretrain_ml.py:636-640 (torch.nn.utils.clip_grad_norm_), and the bf16 working copies of the
// updated weights (row-major and transposed, the operands of the forward / data-gradient GEMMs) written by
// the same kernel that updates the f32 master weights.
//
// All three are HBM-bound streams: 28 B per parameter for the update (+ 4 B for the two bf16 copies instead
// of a separate 8 B/parameter cast pass), 4 B per gradient element for the norm.
#include "common.h"

namespace {

constexpr int SUMSQ_BLOCK_ELEMS = 16384;  // elements per workgroup of the gradient-norm pass

// one partial sum of squares per workgroup, written (not accumulated) at partials[first + blockIdx.x]:
// the final reduction adds them in index order, so the norm is bitwise reproducible
__global__ __launch_bounds__(256) void sumsq_partial_kernel(const float* x, int64_t n, float* partials) {
  const int64_t base = (int64_t)blockIdx.x * SUMSQ_BLOCK_ELEMS;
  const int64_t cnt = min((int64_t)SUMSQ_BLOCK_ELEMS, n - base);
  const float* p = x + base;
  float s = 0.f;
  const bool al = (((uintptr_t)p) & 15) == 0;
  const int64_t nvec = al ? (cnt >> 2) : 0;
  for (int64_t i = threadIdx.x; i < nvec; i += 256) {
    const f32x4 v = reinterpret_cast<const f32x4*>(p)[i];
    s += v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3];
  }
  for (int64_t i = nvec * 4 + threadIdx.x; i < cnt; i += 256) s += p[i] * p[i];
  s = wave_sum(s);
  __shared__ float red[4];
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) partials[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

// norm = sqrt(sum partials); coef = min(1, max_norm / (norm + 1e-6))   (clip_grad_norm_, norm_type 2)
__global__ __launch_bounds__(256) void clip_coef_kernel(const float* partials, int n, float max_norm, float* norm_out,
                                                         float* coef_out) {
  double s = 0.0;
  for (int i = threadIdx.x; i < n; i += 256) s += (double)partials[i];
  __shared__ double red[256];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const float norm = (float)sqrt(red[0]);
    norm_out[0] = norm;
    const float c = max_norm / (norm + 1e-6f);
    coef_out[0] = c < 1.f ? c : 1.f;
  }
}

__device__ __forceinline__ void adam_elem(float& p, float g, float& m, float& v, float b1, float c1, float b2, float c2,
                                          float eps, float step_size, float decay) {
  m = b1 * m + c1 * g;
  v = b2 * v + c2 * g * g;
  p = (p - step_size * (m / (sqrtf(v) + eps))) * decay;
}

__global__ __launch_bounds__(256) void adamw_multi_kernel(const mvptr_adamw_tensor* table, const int32_t* chunk_tensor,
                                                           const int64_t* chunk_offset, int chunk, float b1, float b2,
                                                           float eps, const float* grad_scale) {
  const mvptr_adamw_tensor t = table[chunk_tensor[blockIdx.x]];
  const int64_t off = chunk_offset[blockIdx.x];
  const int64_t cnt = min((int64_t)chunk, t.n - off);
  float* p = t.p + off;
  const float* g = t.g + off;
  float* m = t.m + off;
  float* v = t.v + off;
  const float c1 = 1.f - b1, c2 = 1.f - b2;
  const float gs = grad_scale ? grad_scale[0] : 1.f;
  const bool al = ((((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15) == 0);
  const int64_t nvec = al ? (cnt >> 2) : 0;
  for (int64_t i = threadIdx.x; i < nvec; i += 256) {
    f32x4 pp = reinterpret_cast<f32x4*>(p)[i];
    const f32x4 gg = reinterpret_cast<const f32x4*>(g)[i];
    f32x4 mm = reinterpret_cast<f32x4*>(m)[i];
    f32x4 vv = reinterpret_cast<f32x4*>(v)[i];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float pe = pp[e], me = mm[e], ve = vv[e];
      adam_elem(pe, gg[e] * gs, me, ve, b1, c1, b2, c2, eps, t.step_size, t.decay);
      pp[e] = pe; mm[e] = me; vv[e] = ve;
    }
    reinterpret_cast<f32x4*>(p)[i] = pp;
    reinterpret_cast<f32x4*>(m)[i] = mm;
    reinterpret_cast<f32x4*>(v)[i] = vv;
  }
  for (int64_t i = nvec * 4 + threadIdx.x; i < cnt; i += 256) {
    float pp = p[i], mm = m[i], vv = v[i];
    adam_elem(pp, g[i] * gs, mm, vv, b1, c1, b2, c2, eps, t.step_size, t.decay);
    p[i] = pp;
    m[i] = mm;
    v[i] = vv;
  }
}

// AdamW over 64 x 64 tiles of tensors viewed as [rows, cols], each tile also written as bf16 to the tensor's
// working copies: dst [rows, ld_dst] (columns cols..ld_dst-1 zero: the K padding of the GEMM operands),
// dst_t [cols, ld_dst_t] at column offset col_off_t (transposed copy: operand of the data-gradient GEMMs;
// Q|K|V land side by side in one [H, 3H] matrix), dst_f32 (packed f32 copy: the Q|K|V bias vector).
// Thread t owns the column quad (t & 15) of rows (t >> 4) + 16 k: 16-byte accesses to p, g, m, v, 8-byte
// stores to dst; the bf16 tile goes through LDS for the transposed write (32 B per thread, 128 B per row).
__global__ __launch_bounds__(256) void adamw_mirror_kernel(const mvptr_adamw_mirror_tensor* table, const int* tile_base,
                                                            int n_tensors, float b1, float b2, float eps,
                                                            const float* grad_scale) {
  __shared__ __bf16 tile[64][66];
  int lo = 0, hi = n_tensors - 1;
  const int b = blockIdx.x;
  while (lo < hi) {  // last tensor whose first tile is <= b
    const int mid = (lo + hi + 1) >> 1;
    if (tile_base[mid] <= b) lo = mid; else hi = mid - 1;
  }
  const mvptr_adamw_mirror_tensor t = table[lo];
  const int local = b - tile_base[lo];
  const int wcols = (t.dst != nullptr && t.ld_dst > t.cols) ? (int)t.ld_dst : t.cols;
  const int tiles_x = (wcols + 63) >> 6;
  const int bx = local % tiles_x, by = local / tiles_x;
  const int c0 = bx * 64, r0 = by * 64;
  const int cq = threadIdx.x & 15, rr = threadIdx.x >> 4;
  const int c = c0 + cq * 4;
  const float c1 = 1.f - b1, c2 = 1.f - b2;
  const float gs = grad_scale ? grad_scale[0] : 1.f;
  const bool vec = ((t.cols & 3) == 0) && ((((uintptr_t)t.p | (uintptr_t)t.g | (uintptr_t)t.m | (uintptr_t)t.v | (uintptr_t)t.dst_f32) & 15) == 0);
  __bf16* dst = (__bf16*)t.dst;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int r = r0 + rr + 16 * k;
    float pv[4] = {0.f, 0.f, 0.f, 0.f};
    if (r < t.rows && c < t.cols) {
      const int64_t o = (int64_t)r * t.cols + c;
      if (vec) {
        f32x4 pp = *reinterpret_cast<f32x4*>(t.p + o);
        const f32x4 gg = *reinterpret_cast<const f32x4*>(t.g + o);
        f32x4 mm = *reinterpret_cast<f32x4*>(t.m + o);
        f32x4 vv = *reinterpret_cast<f32x4*>(t.v + o);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float pe = pp[e], me = mm[e], ve = vv[e];
          adam_elem(pe, gg[e] * gs, me, ve, b1, c1, b2, c2, eps, t.step_size, t.decay);
          pp[e] = pe; mm[e] = me; vv[e] = ve;
        }
        *reinterpret_cast<f32x4*>(t.p + o) = pp;
        *reinterpret_cast<f32x4*>(t.m + o) = mm;
        *reinterpret_cast<f32x4*>(t.v + o) = vv;
        if (t.dst_f32 != nullptr) *reinterpret_cast<f32x4*>(t.dst_f32 + o) = pp;
#pragma unroll
        for (int e = 0; e < 4; ++e) pv[e] = pp[e];
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (c + e < t.cols) {
            float pp = t.p[o + e], mm = t.m[o + e], vv = t.v[o + e];
            adam_elem(pp, t.g[o + e] * gs, mm, vv, b1, c1, b2, c2, eps, t.step_size, t.decay);
            t.p[o + e] = pp;
            t.m[o + e] = mm;
            t.v[o + e] = vv;
            if (t.dst_f32 != nullptr) t.dst_f32[o + e] = pp;
            pv[e] = pp;
          }
      }
    }
    bf16x4 o4;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      o4[e] = f2bf(pv[e]);
      tile[rr + 16 * k][cq * 4 + e] = o4[e];
    }
    if (dst != nullptr && r < t.rows) {
      __bf16* dp = dst + (int64_t)r * t.ld_dst + c;
      if (c + 3 < t.ld_dst && ((t.ld_dst & 3) == 0) && (((uintptr_t)dst & 7) == 0)) {
        *reinterpret_cast<bf16x4*>(dp) = o4;
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (c + e < t.ld_dst) dp[e] = o4[e];
      }
    }
  }
  if (t.dst_t == nullptr) return;
  __syncthreads();
  // transposed write: thread -> column ct of the tile (row c0 + ct of dst_t), 16 consecutive source rows
  __bf16* dst_t = (__bf16*)t.dst_t;
  const int ct = threadIdx.x >> 2, rs = (threadIdx.x & 3) * 16;
  const int cc = c0 + ct;
  if (cc >= t.cols) return;
  __bf16* tp = dst_t + (int64_t)cc * t.ld_dst_t + t.col_off_t + r0 + rs;
  if (r0 + rs + 15 < t.rows && ((((uintptr_t)tp) & 15) == 0)) {
    bf16x8 a, bb;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      a[e] = tile[rs + e][ct];
      bb[e] = tile[rs + 8 + e][ct];
    }
    reinterpret_cast<bf16x8*>(tp)[0] = a;
    reinterpret_cast<bf16x8*>(tp)[1] = bb;
  } else {
#pragma unroll
    for (int e = 0; e < 16; ++e)
      if (r0 + rs + e < t.rows) tp[e] = tile[rs + e][ct];
  }
}

}  // namespace

extern "C" int mvptr_adamw_multi(const mvptr_adamw_tensor* table, const int32_t* chunk_tensor,
                                 const int64_t* chunk_offset, int n_chunks, int chunk_elems,
                                 float beta1, float beta2, float eps, const float* grad_scale, void* stream) {
  if (n_chunks <= 0 || chunk_elems <= 0 || (chunk_elems & 3))
    MVPTR_FAIL(MVPTR_BAD_SHAPE, "adamw_multi: n_chunks > 0 and chunk_elems a positive multiple of 4 required");
  if (!table || !chunk_tensor || !chunk_offset) MVPTR_FAIL(MVPTR_BAD_ARG, "adamw_multi: NULL argument");
  hipLaunchKernelGGL(adamw_multi_kernel, dim3(n_chunks), dim3(256), 0, (hipStream_t)stream, table,
                     chunk_tensor, chunk_offset, chunk_elems, beta1, beta2, eps, grad_scale);
  MVPTR_CHECK_LAUNCH("adamw_multi");
  return MVPTR_OK;
}

extern "C" int mvptr_adamw_mirror_multi(const mvptr_adamw_mirror_tensor* table, const int* tile_base, int n_tensors,
                                        int total_tiles, float beta1, float beta2, float eps, const float* grad_scale,
                                        void* stream) {
  if (!table || !tile_base || n_tensors <= 0 || total_tiles <= 0)
    MVPTR_FAIL(MVPTR_BAD_ARG, "adamw_mirror_multi: empty table");
  hipLaunchKernelGGL(adamw_mirror_kernel, dim3(total_tiles), dim3(256), 0, (hipStream_t)stream, table, tile_base,
                     n_tensors, beta1, beta2, eps, grad_scale);
  MVPTR_CHECK_LAUNCH("adamw_mirror_multi");
  return MVPTR_OK;
}

extern "C" int64_t mvptr_sumsq_partials(int64_t n) { return n <= 0 ? 0 : (n + SUMSQ_BLOCK_ELEMS - 1) / SUMSQ_BLOCK_ELEMS; }

extern "C" int mvptr_sumsq_partial(const float* x, int64_t n, float* partials, void* stream) {
  if (!x || !partials || n <= 0) MVPTR_FAIL(MVPTR_BAD_ARG, "sumsq_partial: bad argument");
  const int64_t blocks = mvptr_sumsq_partials(n);
  if (blocks > 0x7fffffff) MVPTR_FAIL(MVPTR_BAD_SHAPE, "sumsq_partial: buffer too large");
  hipLaunchKernelGGL(sumsq_partial_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, n, partials);
  MVPTR_CHECK_LAUNCH("sumsq_partial");
  return MVPTR_OK;
}

extern "C" int mvptr_clip_coef(const float* partials, int n, float max_norm, float* norm_out, float* coef_out, void* stream) {
  if (!partials || !norm_out || !coef_out || n <= 0) MVPTR_FAIL(MVPTR_BAD_ARG, "clip_coef: bad argument");
  hipLaunchKernelGGL(clip_coef_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, partials, n, max_norm, norm_out, coef_out);
  MVPTR_CHECK_LAUNCH("clip_coef");
  return MVPTR_OK;
}
