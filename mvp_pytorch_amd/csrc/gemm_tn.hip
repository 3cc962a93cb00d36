// gemm_tn.hip — dW[N,K] += A[M,N]^T * B[M,K]  (A = dY, B = X; bf16 in, f32 out).
//
// Replaces the weight gradient autograd computes for every nn.Linear on the path
// (addmm backward of modeling_bert.py:348,395,408 and modeling_vlbert.py:71-73 layers).
//
// CDNA4 design: the reduction index (token row m) is the ROW of both row-major operands, so
// the MFMA fragments are fetched with the gfx950 transposed LDS read ds_read_b64_tr_b16 from
// row-major 64x128 tiles (256-B rows, XOR swizzle that is conflict free for these reads).
// 256(n) x 128(k) output tile per 512-thread workgroup, 8 waves as 4x2, each 64x64 = 2x2
// v_mfma_f32_32x32x16_bf16; tiles staged with buffer_load ... lds into a 3-stage ring.
// M is split across blockIdx.y; partial sums are added with f32 atomics whose wave
// instruction covers two 128-B row segments (one 32x32 accumulator register).
#include "common.h"
#include <stdlib.h>

namespace {

constexpr int TN_ = 256, STAGES = 3;
// TM_ token rows per pipeline step: 64 -> 3 x 48 KiB ring (one workgroup per CU),
// 32 -> 3 x 24 KiB ring (two workgroups per CU)

struct GemmTnArgs {
  const __bf16* A;
  const __bf16* B;
  int64_t lda, ldb;
  int M, N, K;
  float* dW;
  int64_t ldw;
  int tiles_n, tiles_k;
  int rows_per_split;
  float* colsum;  // optional: += column sums of A (bias gradient)
  unsigned long long* stamps;  // -DMVPTR_TIMELINE_BUILD only (MVPTR_GEMM_STAMPS)
};

__device__ __forceinline__ int swz256(int row) { return ((row & 3) << 2) | ((row >> 2) & 3); }

__device__ __forceinline__ bf16x8 tr_frag(const char* tile, uint32_t off_lo, uint32_t off_hi) {
  s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
      (__attribute__((address_space(3))) s16x4*)LDS_PTR(tile + off_lo));
  s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
      (__attribute__((address_space(3))) s16x4*)LDS_PTR(tile + off_hi));
  s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(bf16x8, v);
}

extern __shared__ __attribute__((aligned(1024))) char lds[];

// 256(n) x 128(k) output tile per 512-thread workgroup (8 waves as 4(n) x 2(k), each 64x64 =
// 2x2 v_mfma_f32_32x32x16_bf16), 64 token rows per step, 3-stage LDS ring with a counted
// vmcnt(6) + raw s_barrier (two steps in flight).
// TM_ token rows per pipeline step; KSUB = 128-column sub-tiles on the k side of the output tile:
//   KSUB 1: 256(n) x 128(k) tile, waves 4(n) x 2(k), each 64x64   (TM_ 32: 2 workgroups/CU, TM_ 64: 1)
//   KSUB 2: 256(n) x 256(k) tile, waves 2(n) x 4(k), each 128x64  (TM_ 32, 1 workgroup/CU) — a third
//           fewer L2->LDS bytes per FLOP, for outputs with enough tiles to fill the chip
template <int TM_, int KSUB>
__global__ __launch_bounds__(512, ((TM_ == 32 && KSUB == 1) ? 4 : 2)) void gemm_tn_kernel(GemmTnArgs p) {
  constexpr int SUB_B = TM_ * 256;              // one TM_ x 128 bf16 sub-tile
  constexpr int STAGE_B = (2 + KSUB) * SUB_B;   // A = 2 sub-tiles (256 n), B = KSUB sub-tiles
  constexpr int GROUPS = TM_ / 4;               // 4-row wave instructions per sub-tile
  constexpr int NA = 2 * GROUPS / 8, NB = KSUB * GROUPS / 8, NS = TM_ / 16;
  constexpr int TKW = KSUB * 128;               // output tile width (k)
  constexpr int WN = (KSUB == 1) ? 4 : 2, WK = 8 / WN;
  constexpr int NBLK = 256 / (WN * 32);         // 32-row blocks per wave along n
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
#ifdef MVPTR_TIMELINE_BUILD
  unsigned long long tl_start, tl_loop, tl_end;
  asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tl_start)::"memory");
#endif
  const int nt = p.tiles_n * p.tiles_k;
  const int t = xcd_remap(blockIdx.x, nt);
  const int tn = t / p.tiles_k;
  const int tk = t - tn * p.tiles_k;
  const int n0 = tn * TN_, k0 = tk * TKW;
  const int m_begin = blockIdx.y * p.rows_per_split;
  const int m_end = min(p.M, m_begin + p.rows_per_split);
  const int rows = m_end - m_begin;
  if (rows <= 0) return;
  const int ncols = min(TN_, p.N - n0);
  const int kcols = min(TKW, p.K - k0);
  // partial 16-byte chunks at the N/K edge read the row padding (lda/ldb are multiples of 8)
  const int ncols8 = (int)min((int64_t)((ncols + 7) & ~7), p.lda - n0);
  const int kcols8 = (int)min((int64_t)((kcols + 7) & ~7), p.ldb - k0);

  const __amdgpu_buffer_rsrc_t rsA = make_rsrc(
      p.A + (int64_t)m_begin * p.lda + n0, (uint32_t)(((int64_t)(rows - 1) * p.lda + ncols8) * 2));
  const __amdgpu_buffer_rsrc_t rsB = make_rsrc(
      p.B + (int64_t)m_begin * p.ldb + k0, (uint32_t)(((int64_t)(rows - 1) * p.ldb + kcols8) * 2));

  // staging: a wave instruction fills 4 LDS rows (1 KiB) of a TM_ x 128 sub-tile
  uint32_t offA[NA], offB[NB];
#pragma unroll
  for (int i = 0; i < NA; ++i) {
    const int j = i * 8 + wave;  // sub-tile j / GROUPS, row group j % GROUPS
    const int row = (j % GROUPS) * 4 + (lane >> 4);
    const int ch = (lane & 15) ^ swz256(row);
    const int col = (j / GROUPS) * 128 + ch * 8;
    offA[i] = (col < ncols) ? (uint32_t)(row * p.lda * 2 + col * 2) : MVPTR_OOB;
  }
#pragma unroll
  for (int i = 0; i < NB; ++i) {
    const int j = i * 8 + wave;
    const int row = (j % GROUPS) * 4 + (lane >> 4);
    const int ch = (lane & 15) ^ swz256(row);
    const int col = (j / GROUPS) * 128 + ch * 8;
    offB[i] = (col < kcols) ? (uint32_t)(row * p.ldb * 2 + col * 2) : MVPTR_OOB;
  }
  auto stage = [&](int buf, int mrow0) {
    char* la = lds + buf * STAGE_B;
    char* lb = la + 2 * SUB_B;
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      const uint32_t va = (offA[i] == MVPTR_OOB) ? MVPTR_OOB : offA[i] + (uint32_t)(mrow0 * p.lda * 2);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, LDS_PTR(la + (i * 8 + wave) * 1024), 16, va, 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      const uint32_t vb = (offB[i] == MVPTR_OOB) ? MVPTR_OOB : offB[i] + (uint32_t)(mrow0 * p.ldb * 2);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, LDS_PTR(lb + (i * 8 + wave) * 1024), 16, vb, 0, 0, 0);
    }
  };

  const int wn = wave / WK, wk = wave % WK;
  const int g = lane >> 4, i16 = lane & 15;
  const int h = g >> 1, cb = g & 1;
  const int q = i16 >> 2, pp = i16 & 3;
  const int ncol_w = wn * NBLK * 32, kcol_w = wk * 64;     // wave's first n / k column in the tile
  const uint32_t a_sub = (uint32_t)(ncol_w / 128) * SUB_B;  // 128-column sub-tile of A / B
  const uint32_t b_sub = (uint32_t)(kcol_w / 128) * SUB_B;
  uint32_t ta[NS][NBLK][2], tb[NS][2][2];
#pragma unroll
  for (int s = 0; s < NS; ++s)
#pragma unroll
    for (int hl = 0; hl < 2; ++hl) {
      const int row = 16 * s + 8 * h + 4 * hl + q;
#pragma unroll
      for (int b = 0; b < NBLK; ++b) {
        const int cha = (ncol_w % 128) / 8 + b * 4 + 2 * cb + (pp >> 1);
        ta[s][b][hl] = a_sub + row * 256 + ((cha ^ swz256(row)) << 4) + 8 * (pp & 1);
      }
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        const int chb = (kcol_w % 128) / 8 + b * 4 + 2 * cb + (pp >> 1);
        tb[s][b][hl] = b_sub + row * 256 + ((chb ^ swz256(row)) << 4) + 8 * (pp & 1);
      }
    }

  f32x16 acc[NBLK][2];
#pragma unroll
  for (int i = 0; i < NBLK; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  // bias gradient (column sums of A = dY): the waves of the first k-tile that own k-column 0 add up
  // their transposed fragments on the VALU
  const bool do_bias = (p.colsum != nullptr) && (tk == 0) && (wk == 0);
  float bsum[NBLK];
#pragma unroll
  for (int i = 0; i < NBLK; ++i) bsum[i] = 0.f;

  const int nsteps = (rows + TM_ - 1) / TM_;
  stage(0, 0);
  if (nsteps > 1) stage(1, TM_);
  int buf = 0;
  constexpr int LPS = NA + NB;
  for (int st = 0; st < nsteps; ++st) {
    if (st + 1 < nsteps)
      asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(LPS) : "memory");
    else
      asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
    if (st + 2 < nsteps) {
      int nb = buf + 2;
      if (nb >= STAGES) nb -= STAGES;
      stage(nb, (st + 2) * TM_);
    }
    const char* la = lds + buf * STAGE_B;
    const char* lb = la + 2 * SUB_B;
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      bf16x8 af[NBLK], bfr[2];
#pragma unroll
      for (int b = 0; b < NBLK; ++b) af[b] = tr_frag(la, ta[s][b][0], ta[s][b][1]);
#pragma unroll
      for (int b = 0; b < 2; ++b) bfr[b] = tr_frag(lb, tb[s][b][0], tb[s][b][1]);
#pragma unroll
      for (int nb = 0; nb < NBLK; ++nb)
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
          acc[nb][kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[nb], bfr[kb], acc[nb][kb], 0, 0, 0);
      if (do_bias) {  // 4 x v_dot2c_f32_bf16 against (1, 1) per fragment
        const bf16x2 ones = {f2bf(1.f), f2bf(1.f)};
#pragma unroll
        for (int nb = 0; nb < NBLK; ++nb)
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const bf16x2 pr = {af[nb][2 * j], af[nb][2 * j + 1]};
            bsum[nb] = __builtin_amdgcn_fdot2_f32_bf16(pr, ones, bsum[nb], false);
          }
      }
    }
    buf = (buf + 1 == STAGES) ? 0 : buf + 1;
  }

#ifdef MVPTR_TIMELINE_BUILD
  asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tl_loop)::"memory");
#endif
  const int l31 = lane & 31, hh = lane >> 5;
#pragma unroll
  for (int nb = 0; nb < NBLK; ++nb)
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
      const int k = k0 + kcol_w + kb * 32 + l31;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int n = n0 + ncol_w + nb * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
        if (n < p.N && k < p.K) atomicAdd(p.dW + (int64_t)n * p.ldw + k, acc[nb][kb][r]);
      }
    }
  if (do_bias) {
    // lane (l31, hh) holds the partial sum of column n = ncol_w + 32 nb + l31 over its 8 of every 16 rows
#pragma unroll
    for (int nb = 0; nb < NBLK; ++nb) {
      const float tot = bsum[nb] + __shfl_xor(bsum[nb], 32);
      const int n = n0 + ncol_w + nb * 32 + l31;
      if (hh == 0 && n < p.N) atomicAdd(p.colsum + n, tot);
    }
  }
#ifdef MVPTR_TIMELINE_BUILD
  asm volatile("s_waitcnt vmcnt(0)\n\ts_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tl_end)::"memory");
  if (p.stamps != nullptr && tid == 0) {
    unsigned long long* o = p.stamps + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 8;
    o[0] = tl_start;
    o[1] = tl_loop;
    o[2] = tl_end;
  }
#endif
}

__global__ void colsum_kernel(const __bf16* X, int64_t ldx, int M, int N, float* out,
                              int rows_per_block) {
  // block handles a 64-column strip x rows_per_block rows; 256 threads = 4 row lanes x 64 cols
  const int c = blockIdx.x * 64 + (threadIdx.x & 63);
  const int r0 = blockIdx.y * rows_per_block;
  const int r1 = min(M, r0 + rows_per_block);
  float s = 0.f;
  if (c < N)
    for (int r = r0 + (threadIdx.x >> 6); r < r1; r += 4) s += bf2f(X[(int64_t)r * ldx + c]);
  __shared__ float red[256];
  red[threadIdx.x] = s;
  __syncthreads();
  if (threadIdx.x < 64 && c < N) {
    s = red[threadIdx.x] + red[threadIdx.x + 64] + red[threadIdx.x + 128] + red[threadIdx.x + 192];
    atomicAdd(out + c, s);
  }
}

}  // namespace

namespace {

template <int TM_, int KSUB>
int launch_tn(GemmTnArgs a, int splits, hipStream_t stream) {
  const int lds_b = STAGES * (2 + KSUB) * TM_ * 256;
  hipError_t e = hipFuncSetAttribute((const void*)gemm_tn_kernel<TM_, KSUB>,
                                     hipFuncAttributeMaxDynamicSharedMemorySize, lds_b);
  if (e != hipSuccess) MVPTR_FAIL(MVPTR_HIP_ERROR, "gemm_tn: set LDS size: %s", hipGetErrorString(e));
  hipLaunchKernelGGL((gemm_tn_kernel<TM_, KSUB>), dim3(a.tiles_n * a.tiles_k, splits), dim3(512), lds_b, stream, a);
  MVPTR_CHECK_LAUNCH("gemm_tn");
  return MVPTR_OK;
}

struct TnPlan {
  int tm, ksub, splits;
  double cost;
};

// Estimated time of one configuration: whole rounds of workgroups x steps per split, plus the f32
// atomics of every M-split (~1.3 TB/s chip-wide, partly overlapped).
TnPlan plan_tn(int M, int N, int K, int tm, int ksub) {
  const int tiles = ((N + TN_ - 1) / TN_) * ((K + ksub * 128 - 1) / (ksub * 128));
  const int slots = (tm == 32 && ksub == 1) ? 512 : 256;
  // microseconds per 64 token rows of one workgroup (measured on MI355X, round 1)
  const double t64 = (ksub == 2) ? 3.1 : (tm == 64 ? 1.7 : 2.7);
  TnPlan best{tm, ksub, 1, 1e30};
  const int max_splits = (M + 255) / 256;
  for (int sp = 1; sp <= 64 && sp <= max_splits; ++sp) {
    const double rounds = (double)((tiles * sp + slots - 1) / slots);
    const double steps = (double)((M + sp * 64 - 1) / (sp * 64));
    const double cost = rounds * steps * t64 + (double)sp * (double)N * (double)K * 4.0 / 1.3e6 * 0.7;
    if (cost < best.cost) {
      best.cost = cost;
      best.splits = sp;
    }
  }
  return best;
}

}  // namespace

extern "C" int mvptr_gemm_tn(const void* A, int64_t lda, const void* B, int64_t ldb, int M, int N,
                             int K, float* dW, int64_t ldw, float* colsum, void* stream) {
  if (M <= 0 || N <= 0 || K <= 0) MVPTR_FAIL(MVPTR_BAD_SHAPE, "gemm_tn: M,N,K must be > 0");
  if ((lda & 7) || (ldb & 7))
    MVPTR_FAIL(MVPTR_BAD_ALIGN, "gemm_tn: lda, ldb must be multiples of 8");
  if (lda < N || ldb < K) MVPTR_FAIL(MVPTR_BAD_SHAPE, "gemm_tn: lda/ldb smaller than N/K");
  if (((uintptr_t)A & 15) || ((uintptr_t)B & 15))
    MVPTR_FAIL(MVPTR_BAD_ALIGN, "gemm_tn: A and B must be 16-byte aligned");
  // three configurations (see gemm_tn_kernel); pick the cheapest plan.  MVPTR_GEMM_TN = "64" |
  // "32" | "k2" forces one (tuning knob).
  TnPlan plans[3] = {plan_tn(M, N, K, 32, 1), plan_tn(M, N, K, 64, 1), plan_tn(M, N, K, 32, 2)};
  // the 256x256 tile ("k2") measured 10-15 % slower than 256x128 at two workgroups per CU for every
  // shape of this model, so only the first two compete by default
  int pick = (plans[1].cost < plans[0].cost) ? 1 : 0;
  const char* env = getenv("MVPTR_GEMM_TN");
  if (env != nullptr) pick = (env[0] == '6') ? 1 : (env[0] == 'k') ? 2 : 0;
  const TnPlan pl = plans[pick];
  GemmTnArgs a;
  a.A = (const __bf16*)A;
  a.B = (const __bf16*)B;
  a.lda = lda;
  a.ldb = ldb;
  a.M = M;
  a.N = N;
  a.K = K;
  a.dW = dW;
  a.ldw = ldw;
  a.colsum = colsum;
  a.stamps = nullptr;
#ifdef MVPTR_TIMELINE_BUILD
  {
    const char* sp = getenv("MVPTR_GEMM_STAMPS");
    if (sp != nullptr) a.stamps = (unsigned long long*)strtoull(sp, nullptr, 0);
  }
#endif
  a.tiles_n = (N + TN_ - 1) / TN_;
  a.tiles_k = (K + pl.ksub * 128 - 1) / (pl.ksub * 128);
  int rps = (M + pl.splits - 1) / pl.splits;
  rps = (rps + pl.tm - 1) / pl.tm * pl.tm;
  // keep each split's byte span below 2 GiB (buffer offsets are 32-bit)
  const int64_t ldmax = lda > ldb ? lda : ldb;
  while ((int64_t)rps * ldmax * 2 >= (int64_t)0x7fffffff && rps > pl.tm) rps = (rps / 2 + pl.tm - 1) / pl.tm * pl.tm;
  const int splits = (M + rps - 1) / rps;
  a.rows_per_split = rps;
  if (pl.ksub == 2) return launch_tn<32, 2>(a, splits, (hipStream_t)stream);
  if (pl.tm == 64) return launch_tn<64, 1>(a, splits, (hipStream_t)stream);
  return launch_tn<32, 1>(a, splits, (hipStream_t)stream);
}

extern "C" int mvptr_colsum(const void* X, int64_t ldx, int M, int N, float* out, void* stream) {
  if (M <= 0 || N <= 0) MVPTR_FAIL(MVPTR_BAD_SHAPE, "colsum: M,N must be > 0");
  const int rpb = 256;
  dim3 grid((N + 63) / 64, (M + rpb - 1) / rpb);
  hipLaunchKernelGGL(colsum_kernel, grid, dim3(256), 0, (hipStream_t)stream, (const __bf16*)X, ldx,
                     M, N, out, rpb);
  MVPTR_CHECK_LAUNCH("colsum");
  return MVPTR_OK;
}
