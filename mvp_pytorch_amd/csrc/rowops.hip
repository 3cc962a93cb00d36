// rowops.hip — HBM-bound row kernels of the encoder path: LayerNorm fwd/bwd (+dropout),
// embedding gather / scatter, f32<->bf16 cast+pack(+transpose), cross-entropy fwd/bwd.
//
// Reference op sequences replaced (file:line relative to the reference tree):
//   BertLayerNorm.forward            transformers/pytorch_transformers/modeling_bert.py:242-246
//   BertEmbeddings.forward           modeling_bert.py:262-277
//   CrossEntropyLoss(ignore_index=-1) oscar/modeling/modeling_vlbert.py:1228-1251
//   nn.Dropout after LayerNorm / dense modeling_bert.py:276,350,409; modeling_vlbert.py:503
// All are memory-bound: one 64-lane wave per row, 16-byte vector accesses, f32 statistics.
#include "common.h"
#include <string.h>
#include <stdlib.h>

namespace {

constexpr int LN_MAXC = 2;  // 16-byte chunks per lane -> H <= 64*8*2 = 1024

__device__ __forceinline__ int remap_row(int r, int rpg, int gstride, int roff) {
  return (r / rpg) * gstride + roff + (r % rpg);
}

// ------------------------------------------------------------------------------ LayerNorm fwd
__global__ __launch_bounds__(256) void ln_fwd_kernel(const __bf16* z, const float* gamma,
                                                      const float* beta, float eps, __bf16* y,
                                                      float* mean, float* rstd, int M_arg, int H,
                                                      int rpg, int gstride, int roff, DropDev drop, const int* rows_dev) {
  drop = drop_resolve(drop);
  const int M = rows_clamped(M_arg, rows_dev);
  const int lane = threadIdx.x & 63;
  const int nch = H >> 3;
  // gamma / beta of this lane's columns stay in registers across rows
  float gm[LN_MAXC][8], bt[LN_MAXC][8];
#pragma unroll
  for (int c = 0; c < LN_MAXC; ++c)
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int col = (lane + 64 * c) * 8 + e;
      gm[c][e] = (gamma != nullptr && col < H) ? gamma[col] : 1.f;
      bt[c][e] = (gamma != nullptr && col < H) ? beta[col] : 0.f;
    }
  for (int r = blockIdx.x * 4 + (threadIdx.x >> 6); r < M; r += gridDim.x * 4) {
  float v[LN_MAXC][8];
  float s = 0.f;
#pragma unroll
  for (int c = 0; c < LN_MAXC; ++c) {
    const int ch = lane + 64 * c;
    if (ch < nch) {
      const bf16x8 x = *reinterpret_cast<const bf16x8*>(z + (int64_t)r * H + ch * 8);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        v[c][e] = bf2f(x[e]);
        s += v[c][e];
      }
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) v[c][e] = 0.f;
    }
  }
  const float mu = wave_sum(s) / (float)H;
  float q = 0.f;
#pragma unroll
  for (int c = 0; c < LN_MAXC; ++c)
    if (lane + 64 * c < nch)
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float d = v[c][e] - mu;
        q += d * d;
      }
  const float var = wave_sum(q) / (float)H;
  const float rs = 1.0f / sqrtf(var + eps);
  if (lane == 0) {
    if (mean) mean[r] = mu;
    if (rstd) rstd[r] = rs;
  }
  const int64_t orow = remap_row(r, rpg, gstride, roff);
#pragma unroll
  for (int c = 0; c < LN_MAXC; ++c) {
    const int ch = lane + 64 * c;
    if (ch < nch) {
      bf16x8 o;
      float t8[8];
#pragma unroll
      for (int e = 0; e < 8; ++e)
        t8[e] = (gamma != nullptr) ? (v[c][e] - mu) * rs * gm[c][e] + bt[c][e] : v[c][e];
#pragma unroll
      for (int e = 0; e < 8; e += 2)
        drop_apply2(drop, (uint64_t)r * (uint64_t)H + ch * 8 + e, t8[e], t8[e + 1]);
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = f2bf(t8[e]);
      *reinterpret_cast<bf16x8*>(y + orow * H + ch * 8) = o;
    }
  }
  }
}

// ------------------------------------------------------------------------------ LayerNorm bwd
// grid-stride over rows; per-lane column partial sums are reduced across the 4 waves through
// LDS and added with one atomic per column per block.
__global__ __launch_bounds__(256) void ln_bwd_kernel(const __bf16* dy, const __bf16* z,
                                                       const float* mean, const float* rstd,
                                                       const float* gamma, __bf16* dz, __bf16* dd,
                                                       float* partial,
                                                       int M_arg, int H, int rpg, int gstride, int roff,
                                                       DropDev ydrop, DropDev ddrop, const int* rows_dev) {
  ydrop = drop_resolve(ydrop);
  ddrop = drop_resolve(ddrop);
  const int M = rows_clamped(M_arg, rows_dev);
  __shared__ float red[3][3][1024];  // waves 1..3 publish, wave 0 sums
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nch = H >> 3;
  float ag[LN_MAXC][8], ab[LN_MAXC][8], abias[LN_MAXC][8], gm[LN_MAXC][8];
#pragma unroll
  for (int c = 0; c < LN_MAXC; ++c)
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      ag[c][e] = ab[c][e] = abias[c][e] = 0.f;
      const int col = (lane + 64 * c) * 8 + e;
      gm[c][e] = (col < H && gamma != nullptr) ? gamma[col] : 0.f;
    }
  const bool ident = (gamma == nullptr);
  for (int r = blockIdx.x * 4 + wave; r < M; r += gridDim.x * 4) {
    const float mu = ident ? 0.f : mean[r], rs = ident ? 1.f : rstd[r];
    const int64_t irow = remap_row(r, rpg, gstride, roff);
    float xh[LN_MAXC][8], g[LN_MAXC][8];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int c = 0; c < LN_MAXC; ++c) {
      const int ch = lane + 64 * c;
      if (ch < nch) {
        const bf16x8 a = *reinterpret_cast<const bf16x8*>(dy + irow * H + ch * 8);
        const bf16x8 x = *reinterpret_cast<const bf16x8*>(z + (int64_t)r * H + ch * 8);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          float d = bf2f(a[e]);
          d = drop_apply(ydrop, (uint64_t)r * (uint64_t)H + ch * 8 + e, d);
          xh[c][e] = (bf2f(x[e]) - mu) * rs;
          g[c][e] = ident ? d : d * gm[c][e];
          s1 += g[c][e];
          s2 += g[c][e] * xh[c][e];
          ag[c][e] += d * xh[c][e];
          ab[c][e] += d;
        }
      } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) xh[c][e] = g[c][e] = 0.f;
      }
    }
    const float c1 = ident ? 0.f : wave_sum(s1) / (float)H;
    const float c2 = ident ? 0.f : wave_sum(s2) / (float)H;
#pragma unroll
    for (int c = 0; c < LN_MAXC; ++c) {
      const int ch = lane + 64 * c;
      if (ch < nch) {
        bf16x8 o, od;
        float td8[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          td8[e] = rs * (g[c][e] - c1 - xh[c][e] * c2);
          o[e] = f2bf(td8[e]);
        }
#pragma unroll
        for (int e = 0; e < 8; e += 2)
          drop_apply2(ddrop, (uint64_t)r * (uint64_t)H + ch * 8 + e, td8[e], td8[e + 1]);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          od[e] = f2bf(td8[e]);
          abias[c][e] += td8[e];
        }
        *reinterpret_cast<bf16x8*>(dz + (int64_t)r * H + ch * 8) = o;
        if (dd != nullptr) *reinterpret_cast<bf16x8*>(dd + (int64_t)r * H + ch * 8) = od;
      }
    }
  }
  if (wave > 0) {
#pragma unroll
    for (int c = 0; c < LN_MAXC; ++c)
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int col = (lane + 64 * c) * 8 + e;
        red[0][wave - 1][col] = ag[c][e];
        red[1][wave - 1][col] = ab[c][e];
        red[2][wave - 1][col] = abias[c][e];
      }
  }
  __syncthreads();
  if (wave == 0) {
#pragma unroll
    for (int c = 0; c < LN_MAXC; ++c)
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int col = (lane + 64 * c) * 8 + e;
        if (col < H) {
          const float a = ag[c][e] + red[0][0][col] + red[0][1][col] + red[0][2][col];
          const float b = ab[c][e] + red[1][0][col] + red[1][1][col] + red[1][2][col];
          const float d = abias[c][e] + red[2][0][col] + red[2][1][col] + red[2][2][col];
          float* pp = partial + (int64_t)blockIdx.x * 3 * H;
          pp[col] = a;
          pp[H + col] = b;
          pp[2 * H + col] = d;
        }
      }
  }
}

// second stage: out[q][col] += sum over blocks of partial[blk][q][col]; blockIdx.z picks one of up to two LayerNorms
// (an encoder layer finalizes both of its LayerNorm backward passes in one launch: 18 fewer launches per step)
struct LnFinal {
  const float* partial;
  int nblk;
  float *dgamma, *dbeta, *dbias;
};
__global__ __launch_bounds__(256) void ln_bwd_finalize_kernel(LnFinal f0, LnFinal f1, int H) {
  const LnFinal& f = blockIdx.z ? f1 : f0;
  const float* partial = f.partial;
  const int nblk = f.nblk;
  const int i = blockIdx.x * 256 + threadIdx.x;  // over 3*H
  if (i >= 3 * H) return;
  const int q = i / H, col = i - q * H;
  float* dst = (q == 0) ? f.dgamma : (q == 1 ? f.dbeta : f.dbias);
  if (dst == nullptr) return;
  // blockIdx.y slices the partial rows; 32 slices -> 32 adds per address, no contention to speak of
  const int per = (nblk + gridDim.y - 1) / gridDim.y;
  const int b0 = blockIdx.y * per, b1 = min(nblk, b0 + per);
  // eight independent loads in flight per thread: the serial form paid a full memory latency per
  // partial row (21 us per call, 0.9 ms per step)
  float s = 0.f;
  int b = b0;
  for (; b + 8 <= b1; b += 8) {
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = partial[(int64_t)(b + u) * 3 * H + i];
#pragma unroll
    for (int u = 0; u < 8; ++u) s += v[u];
  }
  for (; b < b1; ++b) s += partial[(int64_t)b * 3 * H + i];
  if (b1 > b0) atomicAdd(dst + col, s);
}

// ------------------------------------------------------------ LayerNorm, H = 256 * J fast path
// H = 768 is 1.5 sixteen-byte chunks per lane in the generic kernels above (a quarter of the lane
// slots idle, one or two loads in flight per wave).  Here lane l owns the 4 consecutive columns
// 4l + 256j of each of the J column groups (8-byte accesses, every lane busy, 512 contiguous
// bytes per wave instruction) and a wave keeps RPW rows in flight, so the loads of the next row
// are outstanding while the statistics of the previous one are reduced.
// dropout of the 4 J elements a lane owns of row r in the H = 256 J kernels (columns 256 j + 4 lane ...: two hash pairs per j), all
// 2 J pair hashes in lockstep (common.h drop_pairs); the caller has tested thresh16
template <int J>
__device__ __forceinline__ void row_dropout(const DropDev& d, int r, int lane, bool lo32, float (&t)[4 * J]) {
  uint64_t pr[2 * J];
#pragma unroll
  for (int j = 0; j < J; ++j) {
    const uint64_t idx = (uint64_t)r * (uint64_t)(256 * J) + (uint64_t)(256 * j + 4 * lane);
    pr[2 * j] = idx >> 1;
    pr[2 * j + 1] = (idx >> 1) + 1;
  }
  if (lo32) drop_pairs<2 * J, true>(d, pr, t);
  else drop_pairs<2 * J, false>(d, pr, t);
}

template <int J, int RPW>
__global__ __launch_bounds__(256) void ln_fwd_j_kernel(const __bf16* z, const float* gamma,
                                                        const float* beta, float eps, __bf16* y,
                                                        float* mean, float* rstd, int M_arg, int rpg,
                                                        int gstride, int roff, DropDev drop, const int* rows_dev) {
  drop = drop_resolve(drop);
  const int M = rows_clamped(M_arg, rows_dev);
  constexpr int H = 256 * J;
  const bool lo32 = (uint64_t)M_arg * (uint64_t)H < ((uint64_t)1 << 32);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  f32x4 gm[J], bt[J];
#pragma unroll
  for (int j = 0; j < J; ++j) {
    gm[j] = (gamma != nullptr) ? *reinterpret_cast<const f32x4*>(gamma + 256 * j + 4 * lane) : f32x4{1.f, 1.f, 1.f, 1.f};
    bt[j] = (gamma != nullptr) ? *reinterpret_cast<const f32x4*>(beta + 256 * j + 4 * lane) : f32x4{0.f, 0.f, 0.f, 0.f};
  }
  for (int r0 = (blockIdx.x * 4 + wave) * RPW; r0 < M; r0 += gridDim.x * 4 * RPW) {
    bf16x4 x[RPW][J];
#pragma unroll
    for (int i = 0; i < RPW; ++i) {
      const int r = min(r0 + i, M - 1);
#pragma unroll
      for (int j = 0; j < J; ++j)
        x[i][j] = *reinterpret_cast<const bf16x4*>(z + (int64_t)r * H + 256 * j + 4 * lane);
    }
#pragma unroll
    for (int i = 0; i < RPW; ++i) {
      const int r = r0 + i;
      if (r >= M) break;
      float v[J][4];
      float s = 0.f;
#pragma unroll
      for (int j = 0; j < J; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          v[j][e] = bf2f(x[i][j][e]);
          s += v[j][e];
        }
      // identity mode (gamma == NULL: dropout only) passes the values through untouched
      const float mu = (gamma != nullptr || mean != nullptr) ? wave_sum(s) * (1.0f / (float)H) : 0.f;
      float q = 0.f;
#pragma unroll
      for (int j = 0; j < J; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          v[j][e] -= mu;
          q += v[j][e] * v[j][e];
        }
      const float var = wave_sum(q) * (1.0f / (float)H);
      const float rs = 1.0f / sqrtf(var + eps);
      if (lane == 0) {
        if (mean) mean[r] = mu;
        if (rstd) rstd[r] = rs;
      }
      const int64_t orow = remap_row(r, rpg, gstride, roff);
      float t[4 * J];
#pragma unroll
      for (int j = 0; j < J; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) t[4 * j + e] = (gamma != nullptr) ? v[j][e] * rs * gm[j][e] + bt[j][e] : v[j][e] + mu;
      if (drop.thresh16 != 0) row_dropout<J>(drop, r, lane, lo32, t);
#pragma unroll
      for (int j = 0; j < J; ++j) {
        const bf16x4 o = {f2bf(t[4 * j]), f2bf(t[4 * j + 1]), f2bf(t[4 * j + 2]), f2bf(t[4 * j + 3])};
        *reinterpret_cast<bf16x4*>(y + orow * H + 256 * j + 4 * lane) = o;
      }
    }
  }
}

// One wave keeps TWO sets of RPW rows in registers: the loads of the next set are requested before the current set is reduced and
// stored (PIPE; round 5: the plain loop left the memory pipe idle while a wave computed, 2.3-2.7 TB/s effective in the step).
template <int J, int RPW, bool PIPE>
__global__ __launch_bounds__(256, PIPE ? 2 : 4) void ln_bwd_j_kernel(const __bf16* dy, const __bf16* z,
                                                        const float* mean, const float* rstd,
                                                        const float* gamma, __bf16* dz, __bf16* dd,
                                                        float* partial, int M_arg, int rpg, int gstride,
                                                        int roff, DropDev ydrop, DropDev ddrop, const int* rows_dev) {
  ydrop = drop_resolve(ydrop);
  ddrop = drop_resolve(ddrop);
  const int M = rows_clamped(M_arg, rows_dev);
  constexpr int H = 256 * J;
  const bool lo32 = (uint64_t)M_arg * (uint64_t)H < ((uint64_t)1 << 32);
  __shared__ float red[3][3][H];  // waves 1..3 publish, wave 0 sums
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const bool ident = (gamma == nullptr);
  float ag[J][4], ab[J][4], abias[J][4];
  f32x4 gm[J];
#pragma unroll
  for (int j = 0; j < J; ++j) {
    gm[j] = ident ? f32x4{1.f, 1.f, 1.f, 1.f} : *reinterpret_cast<const f32x4*>(gamma + 256 * j + 4 * lane);
#pragma unroll
    for (int e = 0; e < 4; ++e) ag[j][e] = ab[j][e] = abias[j][e] = 0.f;
  }
  struct RowSet {
    bf16x4 a[RPW][J], x[RPW][J];
    float mu[RPW], rs[RPW];
  };
  auto load_rows = [&](int r0, RowSet& R) {
#pragma unroll
    for (int i = 0; i < RPW; ++i) {
      const int r = min(r0 + i, M - 1);
      const int64_t irow = remap_row(r, rpg, gstride, roff);
#pragma unroll
      for (int j = 0; j < J; ++j) {
        R.a[i][j] = *reinterpret_cast<const bf16x4*>(dy + irow * H + 256 * j + 4 * lane);
        R.x[i][j] = *reinterpret_cast<const bf16x4*>(z + (int64_t)r * H + 256 * j + 4 * lane);
      }
      R.mu[i] = ident ? 0.f : mean[r];
      R.rs[i] = ident ? 1.f : rstd[r];
    }
  };
  auto finish_rows = [&](int r0, const RowSet& R) {
#pragma unroll
    for (int i = 0; i < RPW; ++i) {
      const int r = r0 + i;
      if (r >= M) break;
      float xh[J][4], g[J][4];
      float s1 = 0.f, s2 = 0.f;
      float dall[4 * J];
#pragma unroll
      for (int j = 0; j < J; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) dall[4 * j + e] = bf2f(R.a[i][j][e]);
      if (ydrop.thresh16 != 0) row_dropout<J>(ydrop, r, lane, lo32, dall);
#pragma unroll
      for (int j = 0; j < J; ++j) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float de = dall[4 * j + e];
          xh[j][e] = (bf2f(R.x[i][j][e]) - R.mu[i]) * R.rs[i];
          g[j][e] = de * gm[j][e];
          s1 += g[j][e];
          s2 += g[j][e] * xh[j][e];
          ag[j][e] += de * xh[j][e];
          ab[j][e] += de;
        }
      }
      const float c1 = ident ? 0.f : wave_sum(s1) * (1.0f / (float)H);
      const float c2 = ident ? 0.f : wave_sum(s2) * (1.0f / (float)H);
      float t[4 * J];
#pragma unroll
      for (int j = 0; j < J; ++j) {
#pragma unroll
        for (int e = 0; e < 4; ++e) t[4 * j + e] = R.rs[i] * (g[j][e] - c1 - xh[j][e] * c2);
        const bf16x4 o = {f2bf(t[4 * j]), f2bf(t[4 * j + 1]), f2bf(t[4 * j + 2]), f2bf(t[4 * j + 3])};
        *reinterpret_cast<bf16x4*>(dz + (int64_t)r * H + 256 * j + 4 * lane) = o;
      }
      if (ddrop.thresh16 != 0) row_dropout<J>(ddrop, r, lane, lo32, t);
#pragma unroll
      for (int j = 0; j < J; ++j) {
#pragma unroll
        for (int e = 0; e < 4; ++e) abias[j][e] += t[4 * j + e];
        if (dd != nullptr) {
          const bf16x4 od = {f2bf(t[4 * j]), f2bf(t[4 * j + 1]), f2bf(t[4 * j + 2]), f2bf(t[4 * j + 3])};
          *reinterpret_cast<bf16x4*>(dd + (int64_t)r * H + 256 * j + 4 * lane) = od;
        }
      }
    }
  };
  const int stride = gridDim.x * 4 * RPW;
  int r0 = (blockIdx.x * 4 + wave) * RPW;
  if constexpr (PIPE) {
    RowSet A, B;
    if (r0 < M) load_rows(r0, A);
    while (r0 < M) {
      const int r1 = r0 + stride;
      if (r1 < M) load_rows(r1, B);
      finish_rows(r0, A);
      if (r1 >= M) break;
      const int r2 = r1 + stride;
      if (r2 < M) load_rows(r2, A);
      finish_rows(r1, B);
      r0 = r2;
    }
  } else {
    for (; r0 < M; r0 += stride) {
      RowSet A;
      load_rows(r0, A);
      finish_rows(r0, A);
    }
  }
  if (wave > 0) {
#pragma unroll
    for (int j = 0; j < J; ++j)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int col = 256 * j + 4 * lane + e;
        red[0][wave - 1][col] = ag[j][e];
        red[1][wave - 1][col] = ab[j][e];
        red[2][wave - 1][col] = abias[j][e];
      }
  }
  __syncthreads();
  if (wave == 0) {
    float* pp = partial + (int64_t)blockIdx.x * 3 * H;
#pragma unroll
    for (int j = 0; j < J; ++j)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int col = 256 * j + 4 * lane + e;
        pp[col] = ag[j][e] + red[0][0][col] + red[0][1][col] + red[0][2][col];
        pp[H + col] = ab[j][e] + red[1][0][col] + red[1][1][col] + red[1][2][col];
        pp[2 * H + col] = abias[j][e] + red[2][0][col] + red[2][1][col] + red[2][2][col];
      }
  }
}

// Row statistics of a LayerNorm input from the partial sums its producing GEMM left behind (EPI_RESID_LN: per row and 64-column
// strip the sum and the sum of squares of the stored values): stats[m] = (mean, rstd).  Sums in strip order: reproducible.
__global__ __launch_bounds__(256) void ln_stats_finalize_kernel(const float* part, int strips, int M, float inv_h, float eps, float* stats) {
  const int m = blockIdx.x * 256 + threadIdx.x;
  if (m >= M) return;
  const f32x2* pp = reinterpret_cast<const f32x2*>(part) + (int64_t)m * strips;
  float s1 = 0.f, s2 = 0.f;
  for (int i = 0; i < strips; ++i) {
    const f32x2 v = pp[i];
    s1 += v.x;
    s2 += v.y;
  }
  const float mu = s1 * inv_h;
  const float var = fmaxf(s2 * inv_h - mu * mu, 0.f);
  reinterpret_cast<f32x2*>(stats)[m] = f32x2{mu, 1.0f / sqrtf(var + eps)};
}

// --------------------------------------------------------------------------------- embeddings
__global__ __launch_bounds__(256) void embed_fwd_kernel(const int64_t* ids, const int64_t* pids,
                                                         const int64_t* tids, const float* word,
                                                         const float* pos, const float* type,
                                                         __bf16* z, int rows, int H) {
  const int lane = threadIdx.x & 63;
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= rows) return;
  const float* w = word + ids[r] * (int64_t)H;
  const float* p = pos + pids[r] * (int64_t)H;
  const float* t = type + tids[r] * (int64_t)H;
  for (int c = lane * 4; c < H; c += 256) {
    const f32x4 a = *reinterpret_cast<const f32x4*>(w + c);
    const f32x4 b = *reinterpret_cast<const f32x4*>(p + c);
    const f32x4 d = *reinterpret_cast<const f32x4*>(t + c);
    bf16x4 o = {f2bf(a[0] + b[0] + d[0]), f2bf(a[1] + b[1] + d[1]), f2bf(a[2] + b[2] + d[2]),
                f2bf(a[3] + b[3] + d[3])};
    *reinterpret_cast<bf16x4*>(z + (int64_t)r * H + c) = o;
  }
}

// word (padding_idx 0 skipped, as nn.Embedding(padding_idx=0) does) and position tables
__global__ __launch_bounds__(256) void embed_bwd_kernel(const int64_t* ids, const int64_t* pids,
                                                         const __bf16* dz, float* dword,
                                                         float* dpos, int rows, int H) {
  const int lane = threadIdx.x & 63;
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= rows) return;
  const int64_t id = ids[r];
  float* w = dword + id * (int64_t)H;
  float* p = dpos + pids[r] * (int64_t)H;
  for (int c = lane; c < H; c += 64) {
    const float g = bf2f(dz[(int64_t)r * H + c]);
    if (id != 0) atomicAdd(w + c, g);
    atomicAdd(p + c, g);
  }
}

// token-type table (2 rows): per-block partial sums, then one atomic per (type, column, block)
__global__ __launch_bounds__(256) void embed_type_bwd_kernel(const int64_t* tids, const __bf16* dz,
                                                              float* dtype, int rows, int H,
                                                              int rows_per_block, int ntype) {
  const int c = blockIdx.x * 64 + (threadIdx.x & 63);
  const int sub = threadIdx.x >> 6;
  const int r0 = blockIdx.y * rows_per_block;
  const int r1 = min(rows, r0 + rows_per_block);
  __shared__ float red[4][4][64];
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  if (c < H)
    for (int r = r0 + sub; r < r1; r += 4) {
      const int t = (int)tids[r];
      const float g = bf2f(dz[(int64_t)r * H + c]);
#pragma unroll
      for (int k = 0; k < 4; ++k) acc[k] += (t == k) ? g : 0.f;
    }
#pragma unroll
  for (int k = 0; k < 4; ++k) red[sub][k][threadIdx.x & 63] = acc[k];
  __syncthreads();
  if (sub == 0 && c < H)
    for (int k = 0; k < ntype && k < 4; ++k) {
      const float s = red[0][k][threadIdx.x] + red[1][k][threadIdx.x] + red[2][k][threadIdx.x] + red[3][k][threadIdx.x];
      if (s != 0.f) atomicAdd(dtype + (int64_t)k * H + c, s);
    }
}

// ------------------------------------------------------------------------------------- casts
__global__ __launch_bounds__(256) void cast_pack_kernel(const float* src, int64_t ld_src, int rows,
                                                         int cols, __bf16* dst, int64_t ld_dst,
                                                         __bf16* dst_t, int64_t ld_dst_t,
                                                         int col_off_t) {
  __shared__ float tile[32][33];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
  const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int r = r0 + ty + 8 * k, c = c0 + tx;
    float v = 0.f;
    if (r < rows && c < cols) v = src[(int64_t)r * ld_src + c];
    tile[ty + 8 * k][tx] = v;
    if (dst != nullptr && r < rows && c < ld_dst) dst[(int64_t)r * ld_dst + c] = f2bf(v);
  }
  if (dst_t == nullptr) return;
  __syncthreads();
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int c = c0 + ty + 8 * k, r = r0 + tx;  // transposed: row index of dst_t is c
    if (c < cols && r < rows) dst_t[(int64_t)c * ld_dst_t + col_off_t + r] = f2bf(tile[tx][ty + 8 * k]);
  }
}

// row-major cast without a transposed copy (the 2054-d region features, modeling_vlbert.py:498 input: 105 MB f32 ->
// 53 MB bf16 per 256-pair batch).  Round 6: ONE WAVE PER ROW.  A lane owns the 8-column pieces lane, lane + 64, ... of its row
// (32 bytes of f32 in, one 16-byte store out: a wave instruction covers 2 KiB of contiguous source) and requests ALL of them
// before it converts the first — 8 KiB in flight per wave, 16-byte loads (global loads only need dword alignment: rows of 2054
// floats are 8-byte aligned), no integer division per element.  Rounds 3-5 gave a thread one piece (four 8-byte loads, a 64-bit
// division to find its row) and streamed 3.4 TB/s; columns cols..ld_dst-1 are zero-filled.
template <int PIECES>
__global__ __launch_bounds__(256) void cast_rows_kernel(const float* __restrict__ src, int64_t ld_src, int rows, int cols,
                                                         __bf16* __restrict__ dst, int64_t ld_dst, int chunks) {
  const int lane = threadIdx.x & 63;
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= rows) return;
  const float* sp = src + (int64_t)r * ld_src;
  __bf16* dp = dst + (int64_t)r * ld_dst;
  for (int cb = 0; cb < chunks; cb += 64 * PIECES) {
    f32x4 v[PIECES][2];
#pragma unroll
    for (int j = 0; j < PIECES; ++j) {
      const int c0 = (cb + j * 64 + lane) * 8;
      if (c0 + 7 < cols) {
        v[j][0] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(sp + c0));
        v[j][1] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(sp + c0 + 4));
      } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[j][e >> 2][e & 3] = (c0 + e < cols) ? sp[c0 + e] : 0.f;
      }
    }
#pragma unroll
    for (int j = 0; j < PIECES; ++j) {
      const int c = cb + j * 64 + lane;
      if (c >= chunks) continue;
      bf16x8 o;
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = f2bf(v[j][e >> 2][e & 3]);
      *reinterpret_cast<bf16x8*>(dp + c * 8) = o;
    }
  }
}

// many cast_pack jobs in one launch: block -> (task, 32x32 tile) through the prefix array tile_base
__global__ __launch_bounds__(256) void cast_multi_kernel(const mvptr_cast_task* tasks, const int* tile_base,
                                                          int n_tasks) {
  __shared__ float tile[32][33];
  int lo = 0, hi = n_tasks - 1;
  const int b = blockIdx.x;
  while (lo < hi) {  // last task whose first tile is <= b
    const int mid = (lo + hi + 1) >> 1;
    if (tile_base[mid] <= b) lo = mid; else hi = mid - 1;
  }
  const mvptr_cast_task t = tasks[lo];
  const int local = b - tile_base[lo];
  const int bx = local % t.tiles_x, by = local / t.tiles_x;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
  const int c0 = bx * 32, r0 = by * 32;
  __bf16* dst = (__bf16*)t.dst;
  __bf16* dst_t = (__bf16*)t.dst_t;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int r = r0 + ty + 8 * k, c = c0 + tx;
    float v = 0.f;
    if (r < t.rows && c < t.cols) v = t.src[(int64_t)r * t.ld_src + c];
    tile[ty + 8 * k][tx] = v;
    if (dst != nullptr && r < t.rows && c < t.ld_dst) dst[(int64_t)r * t.ld_dst + c] = f2bf(v);
    if (t.dst_f32 != nullptr && r < t.rows && c < t.cols) t.dst_f32[(int64_t)r * t.cols + c] = v;
  }
  if (dst_t == nullptr) return;
  __syncthreads();
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int c = c0 + ty + 8 * k, r = r0 + tx;
    if (c < t.cols && r < t.rows) dst_t[(int64_t)c * t.ld_dst_t + t.col_off_t + r] = f2bf(tile[tx][ty + 8 * k]);
  }
}

__global__ void cast_f32_kernel(const __bf16* src, int64_t ld_src, int rows, int cols, float* dst,
                                int64_t ld_dst) {
  const int64_t n = (int64_t)rows * cols;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = i / cols, c = i - r * cols;
    dst[r * ld_dst + c] = bf2f(src[r * ld_src + c]);
  }
}

__global__ void dropout_mask_kernel(DropDev d, int64_t n, uint8_t* keep) {
  d = drop_resolve(d);
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    keep[i] = (d.thresh16 == 0 || mvptr_rand16((uint64_t)i, d.seed_lo, d.seed_hi) >= d.thresh16) ? 1 : 0;
}

// ----------------------------------------------------------------------------- cross entropy
__device__ __forceinline__ void block_max_sum(float& m, float& s, float* sm) {
  // combine (max, sum-of-exp) pairs over a 256-thread block
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float m2 = __shfl_xor(m, o), s2 = __shfl_xor(s, o);
    const float mm = fmaxf(m, m2);
    s = s * __expf(m - mm) + s2 * __expf(m2 - mm);
    m = mm;
  }
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (lane == 0) {
    sm[wave] = m;
    sm[4 + wave] = s;
  }
  __syncthreads();
  float mm = fmaxf(fmaxf(sm[0], sm[1]), fmaxf(sm[2], sm[3]));
  float ss = 0.f;
#pragma unroll
  for (int w = 0; w < 4; ++w) ss += sm[4 + w] * __expf(sm[w] - mm);
  m = mm;
  s = ss;
}

__global__ __launch_bounds__(256) void ce_fwd_kernel(const float* logits, int64_t ld,
                                                      const int64_t* labels, float* loss_row,
                                                      float* lse_row, int M, int V) {
  __shared__ float sm[8];
  const int r = blockIdx.x;
  const float* x = logits + (int64_t)r * ld;
  float m = -1e30f, s = 0.f;
  for (int c = threadIdx.x; c < V; c += 256) {
    const float v = x[c];
    const float mm = fmaxf(m, v);
    s = s * __expf(m - mm) + __expf(v - mm);
    m = mm;
  }
  block_max_sum(m, s, sm);
  if (threadIdx.x == 0) {
    const float lse = m + logf(s);
    lse_row[r] = lse;
    const int64_t lab = labels[r];
    loss_row[r] = (lab >= 0 && lab < V) ? (lse - x[lab]) : 0.f;
  }
}

__global__ __launch_bounds__(256) void ce_bwd_kernel(const float* logits, int64_t ld,
                                                      const int64_t* labels, const float* lse_row,
                                                      const float* scale, __bf16* dlogits,
                                                      int64_t ld_d, int M, int V, int Vpad) {
  const int r = blockIdx.x;
  const float* x = logits + (int64_t)r * ld;
  __bf16* d = dlogits + (int64_t)r * ld_d;
  const int64_t lab = labels[r];
  const bool valid = lab >= 0 && lab < V;
  const float sc = valid ? scale[0] : 0.f;
  const float lse = lse_row[r];
  for (int c = threadIdx.x; c < Vpad; c += 256) {
    float g = 0.f;
    if (c < V && valid) g = (__expf(x[c] - lse) - ((int64_t)c == lab ? 1.f : 0.f)) * sc;
    d[c] = f2bf(g);
  }
}

thread_local char g_err[512] = {0};

}  // namespace

void mvptr_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char* mvptr_last_error(void) { return g_err; }

// ---------------------------------------------------------------------------------------------
// dropout salt (common.h drop_resolve): one registered device word per device
namespace {
constexpr int kMaxDev = 64;
const uint32_t* g_drop_salt[kMaxDev] = {nullptr};
}  // namespace
const uint32_t* mvptr_drop_salt() {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDev) return nullptr;
  return g_drop_salt[dev];
}
extern "C" int mvptr_set_dropout_salt(const uint32_t* word) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDev) MVPTR_FAIL(MVPTR_HIP_ERROR, "set_dropout_salt: no current device");
  if (((uintptr_t)word & 3) != 0) MVPTR_FAIL(MVPTR_BAD_ALIGN, "set_dropout_salt: the word must be 4-byte aligned");
  g_drop_salt[dev] = word;
  return MVPTR_OK;
}

// ---------------------------------------------------------------------------------------------
// kernel-configuration knobs (see common.h): constants in the product build, environment + mvptr_set_knob in the
// diagnostic build
#ifndef MVPTR_DIAG_BUILD
const MvptrKnobs& mvptr_knobs() {
  static const MvptrKnobs defaults = {{0}, {0}, 0, 1, 0, 0, {0, 0}, 0ull};
  return defaults;
}
#else
namespace {
MvptrKnobs g_knobs;
bool knob_assign(const char* name, const char* value) {
  const char* v = value ? value : "";
  if (!strcmp(name, "MVPTR_GEMM_CFG")) snprintf(g_knobs.gemm_cfg, sizeof(g_knobs.gemm_cfg), "%s", v);
  else if (!strcmp(name, "MVPTR_GEMM_TN")) snprintf(g_knobs.gemm_tn, sizeof(g_knobs.gemm_tn), "%s", v);
  else if (!strcmp(name, "MVPTR_NT_EXP")) g_knobs.nt_exp = atoi(v);
  else if (!strcmp(name, "MVPTR_TN_GROUP")) g_knobs.tn_group = (v[0] == 0) ? 1 : atoi(v);
  else if (!strcmp(name, "MVPTR_LN_GRID")) g_knobs.ln_grid = atoi(v);
  else if (!strcmp(name, "MVPTR_TN_SPLITS")) g_knobs.tn_splits = atoi(v);
  else if (!strcmp(name, "MVPTR_NT_GROUP")) {
    g_knobs.nt_group[0] = g_knobs.nt_group[1] = 0;
    sscanf(v, "%d,%d", &g_knobs.nt_group[0], &g_knobs.nt_group[1]);
  } else if (!strcmp(name, "MVPTR_GEMM_STAMPS")) g_knobs.stamps = strtoull(v, nullptr, 0);
  else return false;
  return true;
}
bool knobs_from_env() {
  memset(&g_knobs, 0, sizeof(g_knobs));
  g_knobs.tn_group = 1;
  static const char* names[] = {"MVPTR_GEMM_CFG", "MVPTR_GEMM_TN", "MVPTR_NT_EXP", "MVPTR_TN_GROUP",
                                "MVPTR_LN_GRID", "MVPTR_GEMM_STAMPS", "MVPTR_TN_SPLITS", "MVPTR_NT_GROUP"};
  for (const char* n : names) {
    const char* v = getenv(n);
    if (v != nullptr && v[0] != 0) {
      knob_assign(n, v);
      fprintf(stderr, "[mvptr] diagnostic knob %s=%s is active\n", n, v);
    }
  }
  return true;
}
}  // namespace
const MvptrKnobs& mvptr_knobs() {
  static const bool once = knobs_from_env();  // C++11: thread-safe, runs once
  (void)once;
  return g_knobs;
}
// Tools only (A/B of kernel configurations inside one process); not thread-safe against launches.
extern "C" int mvptr_set_knob(const char* name, const char* value) {
  if (!name) MVPTR_FAIL(MVPTR_BAD_ARG, "set_knob: NULL name");
  (void)mvptr_knobs();
  if (!knob_assign(name, value)) MVPTR_FAIL(MVPTR_BAD_ARG, "set_knob: unknown knob '%s'", name);
  return MVPTR_OK;
}
#endif

extern "C" int mvptr_query(int what, int64_t* out) {
  if (!out) MVPTR_FAIL(MVPTR_BAD_ARG, "query: out is NULL");
  if (what == MVPTR_Q_ABI_VERSION) {
    *out = MVPTR_ABI_VERSION;
    return MVPTR_OK;
  }
  int dev = 0;
  hipDeviceProp_t prop;
  if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess)
    MVPTR_FAIL(MVPTR_HIP_ERROR, "query: no HIP device");
  if (what == MVPTR_Q_ARCH_OK) {
    *out = (strncmp(prop.gcnArchName, "gfx950", 6) == 0) ? 1 : 0;
    return MVPTR_OK;
  }
  if (what == MVPTR_Q_NUM_CU) {
    *out = prop.multiProcessorCount;
    return MVPTR_OK;
  }
  MVPTR_FAIL(MVPTR_BAD_ARG, "query: unknown code %d", what);
}

extern "C" int mvptr_layernorm_fwd(const void* z, const float* gamma, const float* beta, float eps,
                                   void* y, float* mean, float* rstd, int M, int H,
                                   int rows_per_group, int group_stride, int row_offset,
                                   const mvptr_dropout* drop, void* stream) {
  return mvptr_layernorm_fwd_rows(z, gamma, beta, eps, y, mean, rstd, M, H, rows_per_group, group_stride, row_offset, drop, nullptr, stream);
}

int mvptr_layernorm_fwd_rows(const void* z, const float* gamma, const float* beta, float eps, void* y, float* mean, float* rstd, int M,
                             int H, int rows_per_group, int group_stride, int row_offset, const mvptr_dropout* drop, const int* rows_dev,
                             void* stream) {
  if (M <= 0) MVPTR_FAIL(MVPTR_BAD_SHAPE, "layernorm_fwd: M must be > 0");
  if ((H & 7) || H > 1024 || H <= 0) MVPTR_FAIL(MVPTR_BAD_SHAPE, "layernorm_fwd: H=%d must be a multiple of 8, <= 1024", H);
  if (rows_per_group <= 0) MVPTR_FAIL(MVPTR_BAD_ARG, "layernorm_fwd: rows_per_group must be > 0");
  if (((uintptr_t)z & 15) || ((uintptr_t)y & 15)) MVPTR_FAIL(MVPTR_BAD_ALIGN, "layernorm_fwd: z,y must be 16-byte aligned");
  if (H == 768 && ((uintptr_t)gamma & 15) == 0 && ((uintptr_t)beta & 15) == 0) {
    constexpr int RPW = 2;
    int g = (M + 4 * RPW - 1) / (4 * RPW);
    if (g > 2048) g = 2048;
    hipLaunchKernelGGL((ln_fwd_j_kernel<3, RPW>), dim3(g), dim3(256), 0, (hipStream_t)stream,
                       (const __bf16*)z, gamma, beta, eps, (__bf16*)y, mean, rstd, M, rows_per_group,
                       group_stride, row_offset, make_dropdev(drop), rows_dev);
    MVPTR_CHECK_LAUNCH("layernorm_fwd");
    return MVPTR_OK;
  }
  int grid = (M + 3) / 4;
  if (grid > 2048) grid = 2048;
  hipLaunchKernelGGL(ln_fwd_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream,
                     (const __bf16*)z, gamma, beta, eps, (__bf16*)y, mean, rstd, M, H,
                     rows_per_group, group_stride, row_offset, make_dropdev(drop), rows_dev);
  MVPTR_CHECK_LAUNCH("layernorm_fwd");
  return MVPTR_OK;
}

static int ln_bwd_grid_cap() {
  const int k = mvptr_knobs().ln_grid;  // tuning knob: partial rows (= blocks) of the backward pass
  int cap = k > 0 ? k : 512;  // 512 partial rows: 50 vs 55 us at M = 38 k (1024), 61 us (256)
  if (cap > 1024) cap = 1024;
  return cap;
}

extern "C" int64_t mvptr_layernorm_bwd_ws_bytes(int M, int H) {
  int grid = (M + 3) / 4;
  if (grid > 1024) grid = 1024;  // workspace sized for the largest grid whatever the knob says
  if (grid < 1) grid = 1;
  return (int64_t)grid * 3 * H * 4;
}

extern "C" int mvptr_layernorm_bwd(const void* dy, const void* z, const float* mean,
                                   const float* rstd, const float* gamma, void* dz, void* dd,
                                   float* dgamma, float* dbeta, float* dbias, int M, int H,
                                   int rows_per_group, int group_stride, int row_offset,
                                   const mvptr_dropout* y_drop, const mvptr_dropout* dense_drop,
                                   void* ws, int64_t ws_bytes, void* stream) {
  return mvptr_layernorm_bwd_rows(dy, z, mean, rstd, gamma, dz, dd, dgamma, dbeta, dbias, M, H, rows_per_group, group_stride, row_offset,
                                  y_drop, dense_drop, ws, ws_bytes, nullptr, stream);
}

extern "C" int mvptr_ln_stats_finalize(const float* row_partials, int M, int H, float eps, float* stats, void* stream) {
  if (!row_partials || !stats || M <= 0 || H <= 0 || (H & 63)) MVPTR_FAIL(MVPTR_BAD_ARG, "ln_stats_finalize: NULL argument, M <= 0 or H %% 64 != 0");
  hipLaunchKernelGGL(ln_stats_finalize_kernel, dim3((M + 255) / 256), dim3(256), 0, (hipStream_t)stream, row_partials, H / 64, M, 1.0f / (float)H,
                     eps, stats);
  MVPTR_CHECK_LAUNCH("ln_stats_finalize");
  return MVPTR_OK;
}

int mvptr_layernorm_bwd_rows(const void* dy, const void* z, const float* mean, const float* rstd, const float* gamma, void* dz, void* dd,
                             float* dgamma, float* dbeta, float* dbias, int M, int H, int rows_per_group, int group_stride, int row_offset,
                             const mvptr_dropout* y_drop, const mvptr_dropout* dense_drop, void* ws, int64_t ws_bytes,
                             const int* rows_dev, void* stream) {
  return mvptr_layernorm_bwd_partial(dy, z, mean, rstd, gamma, dz, dd, dgamma, dbeta, dbias, M, H, rows_per_group, group_stride, row_offset,
                                     y_drop, dense_drop, ws, ws_bytes, rows_dev, nullptr, stream);
}

// `pending` != NULL: the partial rows stay in `ws` and the caller finalizes them later (mvptr_layernorm_bwd_finalize2, with another
// LayerNorm's); NULL: finalized here
int mvptr_layernorm_bwd_partial(const void* dy, const void* z, const float* mean, const float* rstd, const float* gamma, void* dz, void* dd,
                                float* dgamma, float* dbeta, float* dbias, int M, int H, int rows_per_group, int group_stride, int row_offset,
                                const mvptr_dropout* y_drop, const mvptr_dropout* dense_drop, void* ws, int64_t ws_bytes,
                                const int* rows_dev, mvptr_ln_pending* pending, void* stream) {
  if (M <= 0) MVPTR_FAIL(MVPTR_BAD_SHAPE, "layernorm_bwd: M must be > 0");
  if ((H & 7) || H > 1024 || H <= 0) MVPTR_FAIL(MVPTR_BAD_SHAPE, "layernorm_bwd: H=%d must be a multiple of 8, <= 1024", H);
  if (rows_per_group <= 0) MVPTR_FAIL(MVPTR_BAD_ARG, "layernorm_bwd: rows_per_group must be > 0");
  if (!dy || !z || !dz) MVPTR_FAIL(MVPTR_BAD_ARG, "layernorm_bwd: NULL argument");
  if (gamma && (!mean || !rstd)) MVPTR_FAIL(MVPTR_BAD_ARG, "layernorm_bwd: mean/rstd required");
  int grid = (M + 3) / 4;
  if (grid > ln_bwd_grid_cap()) grid = ln_bwd_grid_cap();
  if (!ws || ws_bytes < mvptr_layernorm_bwd_ws_bytes(M, H))
    MVPTR_FAIL(MVPTR_WORKSPACE_TOO_SMALL, "layernorm_bwd: workspace %ld < %ld bytes", (long)ws_bytes,
               (long)mvptr_layernorm_bwd_ws_bytes(M, H));
  if (H == 768 && ((uintptr_t)gamma & 15) == 0) {
    constexpr int RPW = 2;
    int g = (M + 4 * RPW - 1) / (4 * RPW);
    if (g < grid) grid = g;  // every block writes its partial row: the finalize pass reads `grid` of them
    if (mvptr_knobs().nt_exp & (1 << 18))      // diagnostic A/B: the plain load-then-finish loop of rounds 2-4
      hipLaunchKernelGGL((ln_bwd_j_kernel<3, RPW, false>), dim3(grid), dim3(256), 0, (hipStream_t)stream,
                         (const __bf16*)dy, (const __bf16*)z, mean, rstd, gamma, (__bf16*)dz, (__bf16*)dd,
                         (float*)ws, M, rows_per_group, group_stride, row_offset, make_dropdev(y_drop),
                         make_dropdev(dense_drop), rows_dev);
    else if (mvptr_knobs().nt_exp & (1 << 8))      // diagnostic A/B: four rows per set
      hipLaunchKernelGGL((ln_bwd_j_kernel<3, 4, true>), dim3(grid), dim3(256), 0, (hipStream_t)stream,
                         (const __bf16*)dy, (const __bf16*)z, mean, rstd, gamma, (__bf16*)dz, (__bf16*)dd,
                         (float*)ws, M, rows_per_group, group_stride, row_offset, make_dropdev(y_drop),
                         make_dropdev(dense_drop), rows_dev);
    else
      hipLaunchKernelGGL((ln_bwd_j_kernel<3, RPW, true>), dim3(grid), dim3(256), 0, (hipStream_t)stream,
                         (const __bf16*)dy, (const __bf16*)z, mean, rstd, gamma, (__bf16*)dz, (__bf16*)dd,
                         (float*)ws, M, rows_per_group, group_stride, row_offset, make_dropdev(y_drop),
                         make_dropdev(dense_drop), rows_dev);
  } else {
    hipLaunchKernelGGL(ln_bwd_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream,
                       (const __bf16*)dy, (const __bf16*)z, mean, rstd, gamma, (__bf16*)dz,
                       (__bf16*)dd, (float*)ws, M, H, rows_per_group, group_stride,
                       row_offset, make_dropdev(y_drop), make_dropdev(dense_drop), rows_dev);
  }
  MVPTR_CHECK_LAUNCH("layernorm_bwd");
  if (pending != nullptr) {
    pending->partial = (const float*)ws;
    pending->nblk = grid;
    pending->dgamma = dgamma;
    pending->dbeta = dbeta;
    pending->dbias = dbias;
    return MVPTR_OK;
  }
  const LnFinal f{(const float*)ws, grid, dgamma, dbeta, dbias};
  hipLaunchKernelGGL(ln_bwd_finalize_kernel, dim3((3 * H + 255) / 256, 32, 1), dim3(256), 0, (hipStream_t)stream, f, f, H);
  MVPTR_CHECK_LAUNCH("layernorm_bwd_finalize");
  return MVPTR_OK;
}

int mvptr_layernorm_bwd_finalize2(const mvptr_ln_pending* a, const mvptr_ln_pending* b, int H, void* stream) {
  const LnFinal f0{a->partial, a->nblk, a->dgamma, a->dbeta, a->dbias};
  const LnFinal f1{b->partial, b->nblk, b->dgamma, b->dbeta, b->dbias};
  hipLaunchKernelGGL(ln_bwd_finalize_kernel, dim3((3 * H + 255) / 256, 32, 2), dim3(256), 0, (hipStream_t)stream, f0, f1, H);
  MVPTR_CHECK_LAUNCH("layernorm_bwd_finalize2");
  return MVPTR_OK;
}

extern "C" int mvptr_embed_fwd(const int64_t* ids, const int64_t* pos_ids, const int64_t* type_ids,
                               const float* word, const float* pos, const float* type, void* z,
                               int rows, int H, int64_t vocab, int64_t npos, int64_t ntype,
                               void* stream) {
  if (rows <= 0 || (H & 3)) MVPTR_FAIL(MVPTR_BAD_SHAPE, "embed_fwd: rows > 0 and H %% 4 == 0 required");
  if (!ids || !pos_ids || !type_ids) MVPTR_FAIL(MVPTR_BAD_ARG, "embed_fwd: ids/pos_ids/type_ids must be given");
  (void)vocab;
  (void)npos;
  (void)ntype;
  hipLaunchKernelGGL(embed_fwd_kernel, dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, ids,
                     pos_ids, type_ids, word, pos, type, (__bf16*)z, rows, H);
  MVPTR_CHECK_LAUNCH("embed_fwd");
  return MVPTR_OK;
}

extern "C" int mvptr_embed_bwd(const int64_t* ids, const int64_t* pos_ids, const int64_t* type_ids,
                               const void* dz, float* dword, float* dpos, float* dtype, int rows,
                               int H, void* stream) {
  if (rows <= 0) MVPTR_FAIL(MVPTR_BAD_SHAPE, "embed_bwd: rows must be > 0");
  hipLaunchKernelGGL(embed_bwd_kernel, dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, ids,
                     pos_ids, (const __bf16*)dz, dword, dpos, rows, H);
  MVPTR_CHECK_LAUNCH("embed_bwd");
  const int rpb = 512;
  hipLaunchKernelGGL(embed_type_bwd_kernel, dim3((H + 63) / 64, (rows + rpb - 1) / rpb), dim3(256), 0,
                     (hipStream_t)stream, type_ids, (const __bf16*)dz, dtype, rows, H, rpb, 4);
  MVPTR_CHECK_LAUNCH("embed_type_bwd");
  return MVPTR_OK;
}

extern "C" int mvptr_cast_pack(const float* src, int64_t ld_src, int rows, int cols, void* dst,
                               int64_t ld_dst, void* dst_t, int64_t ld_dst_t, int col_off_t,
                               void* stream) {
  if (rows <= 0 || cols <= 0) MVPTR_FAIL(MVPTR_BAD_SHAPE, "cast_pack: rows, cols must be > 0");
  if (dst && ld_dst < cols) MVPTR_FAIL(MVPTR_BAD_SHAPE, "cast_pack: ld_dst < cols");
  const int64_t wcols = (dst && ld_dst > cols) ? ld_dst : cols;
  if (dst && !dst_t && (ld_dst & 7) == 0 && (((uintptr_t)dst) & 15) == 0 && wcols >= 64) {
    // row-major only: the vectorised row cast (whole 16-byte pieces of every destination row, pad columns included)
    const int chunks = (int)(ld_dst / 8);
    const int g = (rows + 3) / 4;
    if ((ld_src & 1) == 0 && (((uintptr_t)src) & 7) == 0 && chunks > 128)
      hipLaunchKernelGGL(cast_rows_kernel<4>, dim3(g), dim3(256), 0, (hipStream_t)stream, src, ld_src, rows, cols, (__bf16*)dst, ld_dst, chunks);
    else
      hipLaunchKernelGGL(cast_rows_kernel<1>, dim3(g), dim3(256), 0, (hipStream_t)stream, src, ld_src, rows, cols, (__bf16*)dst, ld_dst, chunks);
    MVPTR_CHECK_LAUNCH("cast_pack");
    return MVPTR_OK;
  }
  dim3 grid((unsigned)((wcols + 31) / 32), (rows + 31) / 32);
  hipLaunchKernelGGL(cast_pack_kernel, grid, dim3(256), 0, (hipStream_t)stream, src, ld_src, rows,
                     cols, (__bf16*)dst, ld_dst, (__bf16*)dst_t, ld_dst_t, col_off_t);
  MVPTR_CHECK_LAUNCH("cast_pack");
  return MVPTR_OK;
}

extern "C" int mvptr_cast_multi(const mvptr_cast_task* tasks, const int* tile_base, int n_tasks,
                                int total_tiles, void* stream) {
  if (!tasks || !tile_base || n_tasks <= 0 || total_tiles <= 0)
    MVPTR_FAIL(MVPTR_BAD_ARG, "cast_multi: empty task table");
  hipLaunchKernelGGL(cast_multi_kernel, dim3(total_tiles), dim3(256), 0, (hipStream_t)stream, tasks,
                     tile_base, n_tasks);
  MVPTR_CHECK_LAUNCH("cast_multi");
  return MVPTR_OK;
}

extern "C" int mvptr_cast_f32(const void* src, int64_t ld_src, int rows, int cols, float* dst,
                              int64_t ld_dst, void* stream) {
  if (rows <= 0 || cols <= 0) MVPTR_FAIL(MVPTR_BAD_SHAPE, "cast_f32: rows, cols must be > 0");
  const int64_t n = (int64_t)rows * cols;
  int grid = (int)((n + 255) / 256);
  if (grid > 4096) grid = 4096;
  hipLaunchKernelGGL(cast_f32_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream,
                     (const __bf16*)src, ld_src, rows, cols, dst, ld_dst);
  MVPTR_CHECK_LAUNCH("cast_f32");
  return MVPTR_OK;
}

extern "C" int mvptr_dropout_mask(const mvptr_dropout* drop, int64_t n, uint8_t* keep, void* stream) {
  if (n <= 0) MVPTR_FAIL(MVPTR_BAD_SHAPE, "dropout_mask: n must be > 0");
  int grid = (int)((n + 255) / 256);
  if (grid > 4096) grid = 4096;
  hipLaunchKernelGGL(dropout_mask_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream,
                     make_dropdev(drop), n, keep);
  MVPTR_CHECK_LAUNCH("dropout_mask");
  return MVPTR_OK;
}

extern "C" int mvptr_ce_fwd(const float* logits, int64_t ld, const int64_t* labels, float* loss_row,
                            float* lse_row, int M, int V, void* stream) {
  if (M <= 0 || V <= 0) MVPTR_FAIL(MVPTR_BAD_SHAPE, "ce_fwd: M, V must be > 0");
  hipLaunchKernelGGL(ce_fwd_kernel, dim3(M), dim3(256), 0, (hipStream_t)stream, logits, ld, labels,
                     loss_row, lse_row, M, V);
  MVPTR_CHECK_LAUNCH("ce_fwd");
  return MVPTR_OK;
}

extern "C" int mvptr_ce_bwd(const float* logits, int64_t ld, const int64_t* labels,
                            const float* lse_row, const float* scale, void* dlogits, int64_t ld_d,
                            int M, int V, int Vpad, void* stream) {
  if (M <= 0 || V <= 0 || Vpad < V || ld_d < Vpad) MVPTR_FAIL(MVPTR_BAD_SHAPE, "ce_bwd: bad shape");
  hipLaunchKernelGGL(ce_bwd_kernel, dim3(M), dim3(256), 0, (hipStream_t)stream, logits, ld, labels,
                     lse_row, scale, (__bf16*)dlogits, ld_d, M, V, Vpad);
  MVPTR_CHECK_LAUNCH("ce_bwd");
  return MVPTR_OK;
}

