// wra.hip — the weakly-supervised phrase / region alignment loss of the pre-training step on rows tapped from the
// packed joint output.
//
// Replaces oscar/modeling/modeling_vlbert.py:1285-1300 (phrase_mod == 'sample') with get_pos_neg_sims (:1553-1596)
// and t2i_sim (:1543-1550): per sample the phrase rows and the region rows of the joint sequence are L2-normalised,
// every phrase takes one of its three most similar regions (a drawn rank 0..2) in the sample's own image and in one
// other image of the batch, the two means over the phrases enter a hinge with margin 0.2, and the loss is the mean
// hinge over the samples that have phrases.  The reference walks the samples in Python and the torch restatement is
// ~150 small launches forward + backward; here: one workgroup per sample, two launches forward and one backward.
// f32 arithmetic on bf16 rows, sums in a fixed order (no atomics): run-to-run identical.
#include "common.h"

namespace {

constexpr int WRA_THREADS = 1024;
constexpr int WRA_WAVES = WRA_THREADS / 64;
constexpr int WRA_MAXK = 8;  // column pairs per lane: H <= 1024

struct WraArgs {
  const __bf16* txt;  // [n, Pw, H] phrase rows (row k of sample i = phrase k; rows >= the phrase count are ignored)
  const __bf16* reg;  // [n, Rw, H] region rows
  const int64_t* phrase_index;  // [n, 2]
  const int64_t* img_index;     // [n, 2]
  const int64_t* pos_pick;      // [n, Pw] drawn rank 0..2 in the own image
  const int64_t* neg_pick;      // [n, Pw] drawn rank in the other image
  const int64_t* neg_img;       // [n] the other image
  int n, Pw, Rw, H;
  float* hinge;    // [n]
  int32_t* cnt;    // [n, 2] phrases, regions
  int32_t* sel;    // [n, Pw, 2] chosen region (own image, other image); -1: fewer regions than the rank
  float* sval;     // [n, Pw, 2] its cosine similarity
  float* inv_p;    // [n, Pw] 1 / max(|x|, 1e-12) of the phrase rows
  float* inv_r;    // [n, Rw] of the region rows
};

__device__ __forceinline__ int clampi(int64_t v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : (int)v); }

// this lane's column pairs of one bf16 row, as f32
__device__ __forceinline__ void load_row(const __bf16* row, int H, int lane, float (&v)[2 * WRA_MAXK]) {
#pragma unroll
  for (int k = 0; k < WRA_MAXK; ++k) {
    const int c = 2 * lane + 128 * k;
    if (c < H) {
      const bf16x2 x = *reinterpret_cast<const bf16x2*>(row + c);
      v[2 * k] = bf2f(x.x);
      v[2 * k + 1] = bf2f(x.y);
    } else {
      v[2 * k] = 0.f;
      v[2 * k + 1] = 0.f;
    }
  }
}

__device__ __forceinline__ float dot_rows(const float (&a)[2 * WRA_MAXK], const float (&b)[2 * WRA_MAXK]) {
  float s = 0.f;
#pragma unroll
  for (int k = 0; k < 2 * WRA_MAXK; ++k) s = fmaf(a[k], b[k], s);
  return wave_sum(s);
}

__device__ __forceinline__ float inv_norm(float ss) { return 1.0f / fmaxf(sqrtf(ss), 1e-12f); }  // F.normalize(p=2, eps=1e-12)

__global__ __launch_bounds__(WRA_THREADS) void wra_fwd_kernel(WraArgs a) {
  extern __shared__ __align__(16) unsigned char smem[];
  const int i = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int Pw = a.Pw, Rw = a.Rw, H = a.H;
  __bf16* ph = reinterpret_cast<__bf16*>(smem);                                   // [Pw, H]
  float* sims = reinterpret_cast<float*>(smem + (size_t)Pw * H * sizeof(__bf16));  // [2, Pw, Rw]
  float* invp = sims + 2 * Pw * Rw;                                               // [Pw]
  float* picked = invp + Pw;                                                      // [2, Pw]
  const int np = clampi(a.phrase_index[2 * i + 1] - a.phrase_index[2 * i], 0, Pw);
  const int nr = clampi(a.img_index[2 * i + 1] - a.img_index[2 * i], 0, Rw);
  const int m = clampi(a.neg_img[i], 0, a.n - 1);
  const int nrm = np ? clampi(a.img_index[2 * m + 1] - a.img_index[2 * m], 0, Rw) : 0;
  if (tid == 0) {
    a.cnt[2 * i] = np;
    a.cnt[2 * i + 1] = nr;
  }
  // phrase rows -> LDS, their norms
  const __bf16* txt = a.txt + (size_t)i * Pw * H;
  for (int e = tid; e < np * (H / 2); e += WRA_THREADS)
    reinterpret_cast<uint32_t*>(ph)[e] = reinterpret_cast<const uint32_t*>(txt)[e];
  __syncthreads();
  float t[2 * WRA_MAXK], v[2 * WRA_MAXK];
  for (int p = wave; p < np; p += WRA_WAVES) {
    load_row(ph + (size_t)p * H, H, lane, t);
    const float ip = inv_norm(dot_rows(t, t));
    if (lane == 0) {
      invp[p] = ip;
      a.inv_p[(size_t)i * Pw + p] = ip;
    }
  }
  __syncthreads();
  // one wave per region row (own image first, then the other image): its norm, its similarity to every phrase
  for (int q = wave; q < nr + nrm; q += WRA_WAVES) {
    const int side = q >= nr, r = side ? q - nr : q;
    load_row(a.reg + ((size_t)(side ? m : i) * Rw + r) * H, H, lane, v);
    const float ir = inv_norm(dot_rows(v, v));
    if (!side && lane == 0) a.inv_r[(size_t)i * Rw + r] = ir;
    for (int p = 0; p < np; ++p) {
      load_row(ph + (size_t)p * H, H, lane, t);
      const float s = dot_rows(t, v) * invp[p] * ir;
      if (lane == 0) sims[((size_t)side * Pw + p) * Rw + r] = s;
    }
  }
  __syncthreads();
  // top 3 of each phrase's row, the drawn rank
  if (tid < 2 * Pw) {
    const int side = tid / Pw, p = tid % Pw;
    if (p < np) {
      const float* row = sims + ((size_t)side * Pw + p) * Rw;
      const int cnt = side ? nrm : nr;
      float v0 = -INFINITY, v1 = -INFINITY, v2 = -INFINITY;
      int i0 = -1, i1 = -1, i2 = -1;
      for (int r = 0; r < cnt; ++r) {
        const float s = row[r];
        if (s > v0) {
          v2 = v1, i2 = i1, v1 = v0, i1 = i0, v0 = s, i0 = r;
        } else if (s > v1) {
          v2 = v1, i2 = i1, v1 = s, i1 = r;
        } else if (s > v2) {
          v2 = s, i2 = r;
        }
      }
      // The reference's topk(3) needs >= 3 valid regions per image (vl:1547 raises otherwise; the Python side asserts it
      // on the device).  An image with fewer must not leak -inf into the loss: the drawn rank is clamped to the
      // regions that exist, and an image without any contributes a similarity of 0 and no gradient (sel = -1).
      const int k = min(clampi((side ? a.neg_pick : a.pos_pick)[(size_t)i * Pw + p], 0, 2), max(cnt, 1) - 1);
      const float pv = cnt == 0 ? 0.f : (k == 0 ? v0 : (k == 1 ? v1 : v2));
      const int pi = cnt == 0 ? -1 : (k == 0 ? i0 : (k == 1 ? i1 : i2));
      picked[side * Pw + p] = pv;
      a.sel[((size_t)i * Pw + p) * 2 + side] = pi;
      a.sval[((size_t)i * Pw + p) * 2 + side] = pv;
    }
  }
  __syncthreads();
  if (tid == 0) {
    float h = 0.f;
    if (np) {
      float pos = 0.f, neg = 0.f;
      for (int p = 0; p < np; ++p) pos += picked[p], neg += picked[Pw + p];
      h = fmaxf(neg / (float)np + 0.2f - pos / (float)np, 0.f);
    }
    a.hinge[i] = h;
  }
}

// loss = sum of the hinges / samples that have phrases (vl:1297-1300); coef[i] = d loss / d (a picked similarity of
// sample i's own image) up to sign: 1 / (samples * phrases of i) where the hinge is active, else 0.
__global__ __launch_bounds__(256) void wra_reduce_kernel(const float* hinge, const int32_t* cnt, int n, float* loss, float* coef) {
  __shared__ float sum_s[256];
  __shared__ int val_s[256];
  const int tid = threadIdx.x;
  float s = 0.f;
  int nv = 0;
  for (int i = tid; i < n; i += 256) {
    if (cnt[2 * i] > 0) s += hinge[i], ++nv;
  }
  sum_s[tid] = s;
  val_s[tid] = nv;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (tid < o) sum_s[tid] += sum_s[tid + o], val_s[tid] += val_s[tid + o];
    __syncthreads();
  }
  const float nvalid = (float)val_s[0];
  if (tid == 0) loss[0] = sum_s[0] / nvalid;
  for (int i = tid; i < n; i += 256) coef[i] = (cnt[2 * i] > 0 && hinge[i] > 0.f) ? 1.0f / (nvalid * (float)cnt[2 * i]) : 0.f;
}

struct WraBwdArgs {
  const __bf16* txt;
  const __bf16* reg;
  const int64_t* neg_img;
  int n, Pw, Rw, H;
  const int32_t* cnt;
  const int32_t* sel;
  const float* sval;
  const float* inv_p;
  const float* inv_r;
  const float* coef;
  const float* gout;  // d / d loss (one f32 on the device)
  __bf16* d_txt;      // [n, Pw, H]
  __bf16* d_reg;      // [n, Rw, H]
};

__device__ __forceinline__ void store_row(__bf16* row, int H, int lane, const float (&v)[2 * WRA_MAXK], float scale) {
#pragma unroll
  for (int k = 0; k < WRA_MAXK; ++k) {
    const int c = 2 * lane + 128 * k;
    if (c < H) {
      bf16x2 o;
      o.x = f2bf(v[2 * k] * scale);
      o.y = f2bf(v[2 * k + 1] * scale);
      *reinterpret_cast<bf16x2*>(row + c) = o;
    }
  }
}

// acc += c * (x * ix - v * s): the gradient of c * <x * ix, v> with respect to v's un-normalised row, before the
// final 1 / |row| (v is unit length, s = <x * ix, v>)
__device__ __forceinline__ void add_projected(float (&acc)[2 * WRA_MAXK], const float (&x)[2 * WRA_MAXK], float ix,
                                              const float (&v)[2 * WRA_MAXK], float s, float c) {
#pragma unroll
  for (int k = 0; k < 2 * WRA_MAXK; ++k) acc[k] = fmaf(c, fmaf(x[k], ix, -v[k] * s), acc[k]);
}

// Workgroup j writes the gradient of sample j's phrase rows and of image j's region rows.  A region row collects, in a
// fixed order, the phrases of sample j that chose it and the phrases of every sample that drew image j as its other
// image and chose it there.
__global__ __launch_bounds__(WRA_THREADS) void wra_bwd_kernel(WraBwdArgs a) {
  extern __shared__ __align__(16) unsigned char smem[];
  int* match = reinterpret_cast<int*>(smem);  // samples (ascending) whose other image is j and whose hinge is active
  __shared__ int n_match_s;
  const int j = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int Pw = a.Pw, Rw = a.Rw, H = a.H;
  const float g = a.gout[0];
  const int np = a.cnt[2 * j], nr = a.cnt[2 * j + 1];
  const float cj = a.coef[j] * g;
  if (wave == 0) {
    int count = 0;
    for (int base = 0; base < a.n; base += 64) {
      const int i = base + lane;
      const bool hit = i < a.n && clampi(a.neg_img[i], 0, a.n - 1) == j && a.coef[i] != 0.f;
      const uint64_t mask = __ballot(hit);
      if (hit) match[count + __popcll(mask & ((1ull << lane) - 1))] = i;
      count += __popcll(mask);
    }
    if (lane == 0) n_match_s = count;
  }
  float x[2 * WRA_MAXK], v[2 * WRA_MAXK], acc[2 * WRA_MAXK];
  // phrase rows: d x_p = c / |x_p| * ((u - v) - t_p * (s_neg - s_pos))
  const int m = clampi(a.neg_img[j], 0, a.n - 1);
  for (int p = wave; p < Pw; p += WRA_WAVES) {
#pragma unroll
    for (int k = 0; k < 2 * WRA_MAXK; ++k) acc[k] = 0.f;
    float ip = 0.f;
    if (p < np && cj != 0.f) {
      const size_t e = ((size_t)j * Pw + p) * 2;
      ip = a.inv_p[(size_t)j * Pw + p];
      load_row(a.txt + ((size_t)j * Pw + p) * H, H, lane, x);
      const int r_pos = a.sel[e], r_neg = a.sel[e + 1];
      if (r_pos >= 0) {
        load_row(a.reg + ((size_t)j * Rw + r_pos) * H, H, lane, v);
        const float ir = a.inv_r[(size_t)j * Rw + r_pos], s = a.sval[e];
#pragma unroll
        for (int k = 0; k < 2 * WRA_MAXK; ++k) acc[k] -= cj * (v[k] * ir - x[k] * ip * s);
      }
      if (r_neg >= 0) {
        load_row(a.reg + ((size_t)m * Rw + r_neg) * H, H, lane, v);
        const float ir = a.inv_r[(size_t)m * Rw + r_neg], s = a.sval[e + 1];
#pragma unroll
        for (int k = 0; k < 2 * WRA_MAXK; ++k) acc[k] += cj * (v[k] * ir - x[k] * ip * s);
      }
    }
    store_row(a.d_txt + ((size_t)j * Pw + p) * H, H, lane, acc, ip);
  }
  __syncthreads();
  const int n_match = n_match_s;
  for (int r = wave; r < Rw; r += WRA_WAVES) {
#pragma unroll
    for (int k = 0; k < 2 * WRA_MAXK; ++k) acc[k] = 0.f;
    float ir = 0.f;
    if (r < nr && (cj != 0.f || n_match)) {
      ir = a.inv_r[(size_t)j * Rw + r];
      load_row(a.reg + ((size_t)j * Rw + r) * H, H, lane, v);
#pragma unroll
      for (int k = 0; k < 2 * WRA_MAXK; ++k) v[k] *= ir;
      for (int q = (cj != 0.f ? -1 : 0); q < n_match; ++q) {
        const int i = q < 0 ? j : match[q];
        const int side = q < 0 ? 0 : 1;
        const float c = q < 0 ? -cj : a.coef[i] * g;
        const int npi = a.cnt[2 * i];
        for (int p0 = 0; p0 < npi; p0 += 64) {
          const int p = p0 + lane;
          const bool hit = p < npi && a.sel[((size_t)i * Pw + p) * 2 + side] == r;
          uint64_t mask = __ballot(hit);
          while (mask) {
            const int pp = p0 + __builtin_ctzll(mask);
            mask &= mask - 1;
            load_row(a.txt + ((size_t)i * Pw + pp) * H, H, lane, x);
            add_projected(acc, x, a.inv_p[(size_t)i * Pw + pp], v, a.sval[((size_t)i * Pw + pp) * 2 + side], c);
          }
        }
      }
    }
    store_row(a.d_reg + ((size_t)j * Rw + r) * H, H, lane, acc, ir);
  }
}

// rows_p[i, k] = packed row of phrase k of sample i (slot p0 + k of the joint sequence), rows_r[i, k] = of region k
// (slot i0 + k); -1 beyond the sample's counts.  pos: [>= n, Lj] packed row of every slot (rows.hip), -1 = padded.
__global__ __launch_bounds__(256) void wra_rows_kernel(const int32_t* pos, int Lj, const int64_t* phrase_index,
                                                       const int64_t* img_index, int n, int Pw, int Rw, int32_t* rows_p,
                                                       int32_t* rows_r) {
  const int i = blockIdx.x;
  const int64_t p0 = phrase_index[2 * i], p1 = phrase_index[2 * i + 1], i0 = img_index[2 * i], i1 = img_index[2 * i + 1];
  for (int k = threadIdx.x; k < Pw + Rw; k += 256) {
    const bool ph = k < Pw;
    const int64_t kk = ph ? k : k - Pw;
    const int64_t slot = (ph ? p0 : i0) + kk;
    const bool ok = kk < (ph ? p1 - p0 : i1 - i0) && slot >= 0 && slot < Lj;
    const int32_t row = ok ? pos[(size_t)i * Lj + slot] : -1;
    if (ph) rows_p[(size_t)i * Pw + kk] = row;
    else rows_r[(size_t)i * Rw + kk] = row;
  }
}

size_t wra_fwd_lds(int Pw, int Rw, int H) {
  return (size_t)Pw * H * sizeof(__bf16) + ((size_t)2 * Pw * Rw + 3 * (size_t)Pw) * sizeof(float);
}

}  // namespace

extern "C" int mvptr_wra_rows(const int32_t* pos, int Lj, const int64_t* phrase_index, const int64_t* img_index, int n, int Pw,
                              int Rw, int32_t* rows_p, int32_t* rows_r, void* stream) {
  if (!pos || !phrase_index || !img_index || !rows_p || !rows_r) MVPTR_FAIL(MVPTR_BAD_ARG, "wra_rows: NULL pointer");
  if (n <= 0 || Lj <= 0 || Pw <= 0 || Rw <= 0) MVPTR_FAIL(MVPTR_BAD_SHAPE, "wra_rows: n, Lj, Pw, Rw must be positive");
  hipLaunchKernelGGL(wra_rows_kernel, dim3(n), dim3(256), 0, (hipStream_t)stream, pos, Lj, phrase_index, img_index, n, Pw, Rw,
                     rows_p, rows_r);
  MVPTR_CHECK_LAUNCH("wra_rows");
  return MVPTR_OK;
}

extern "C" int mvptr_wra_fwd(const void* txt, const void* reg, const int64_t* phrase_index, const int64_t* img_index,
                             const int64_t* pos_pick, const int64_t* neg_pick, const int64_t* neg_img, int n, int Pw, int Rw,
                             int H, float* loss, float* hinge, float* coef, int32_t* cnt, int32_t* sel, float* sval,
                             float* inv_p, float* inv_r, void* stream) {
  if (!txt || !reg || !phrase_index || !img_index || !pos_pick || !neg_pick || !neg_img || !loss || !hinge || !coef || !cnt ||
      !sel || !sval || !inv_p || !inv_r)
    MVPTR_FAIL(MVPTR_BAD_ARG, "wra_fwd: NULL pointer");
  if (n <= 0 || Pw <= 0 || Rw <= 0 || H <= 0 || (H & 1) || H > 128 * WRA_MAXK || 2 * Pw > WRA_THREADS)
    MVPTR_FAIL(MVPTR_BAD_SHAPE, "wra_fwd: need even H <= %d, Pw <= %d (got n=%d Pw=%d Rw=%d H=%d)", 128 * WRA_MAXK,
               WRA_THREADS / 2, n, Pw, Rw, H);
  if (((uintptr_t)txt & 3) || ((uintptr_t)reg & 3)) MVPTR_FAIL(MVPTR_BAD_ALIGN, "wra_fwd: rows must be 4-byte aligned");
  const size_t lds = wra_fwd_lds(Pw, Rw, H);
  if (lds > 150 * 1024) MVPTR_FAIL(MVPTR_BAD_SHAPE, "wra_fwd: phrase grid %d x %d with %d regions needs %zu bytes of LDS", Pw, H, Rw, lds);
  // per call, like the attention kernels: the attribute belongs to the current device's copy of the kernel
  if (hipFuncSetAttribute((const void*)wra_fwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
    MVPTR_FAIL(MVPTR_HIP_ERROR, "wra_fwd: cannot raise the dynamic LDS limit");
  WraArgs a{(const __bf16*)txt, (const __bf16*)reg, phrase_index, img_index, pos_pick, neg_pick, neg_img, n, Pw, Rw, H,
            hinge, cnt, sel, sval, inv_p, inv_r};
  hipLaunchKernelGGL(wra_fwd_kernel, dim3(n), dim3(WRA_THREADS), lds, (hipStream_t)stream, a);
  hipLaunchKernelGGL(wra_reduce_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, hinge, cnt, n, loss, coef);
  MVPTR_CHECK_LAUNCH("wra_fwd");
  return MVPTR_OK;
}

extern "C" int mvptr_wra_bwd(const void* txt, const void* reg, const int64_t* neg_img, int n, int Pw, int Rw, int H,
                             const int32_t* cnt, const int32_t* sel, const float* sval, const float* inv_p, const float* inv_r,
                             const float* coef, const float* gout, void* d_txt, void* d_reg, void* stream) {
  if (!txt || !reg || !neg_img || !cnt || !sel || !sval || !inv_p || !inv_r || !coef || !gout || !d_txt || !d_reg)
    MVPTR_FAIL(MVPTR_BAD_ARG, "wra_bwd: NULL pointer");
  if (n <= 0 || Pw <= 0 || Rw <= 0 || H <= 0 || (H & 1) || H > 128 * WRA_MAXK || (size_t)n * sizeof(int) > 64 * 1024)
    MVPTR_FAIL(MVPTR_BAD_SHAPE, "wra_bwd: need even H <= %d and n <= 16384 (got n=%d Pw=%d Rw=%d H=%d)", 128 * WRA_MAXK, n, Pw, Rw, H);
  WraBwdArgs a{(const __bf16*)txt, (const __bf16*)reg, neg_img, n, Pw, Rw, H, cnt, sel, sval, inv_p, inv_r, coef, gout,
               (__bf16*)d_txt, (__bf16*)d_reg};
  hipLaunchKernelGGL(wra_bwd_kernel, dim3(n), dim3(WRA_THREADS), (size_t)n * sizeof(int), (hipStream_t)stream, a);
  MVPTR_CHECK_LAUNCH("wra_bwd");
  return MVPTR_OK;
}
