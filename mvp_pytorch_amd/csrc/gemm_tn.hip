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
#include <string.h>
#include <stddef.h>
#include <type_traits>
#include <utility>

namespace {

constexpr int TN_ = 256;
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
  int order_n_major;  // A/B knob (MVPTR_NT_EXP bit 5): n-tile-major tile order whatever the grid shape
  unsigned long long* stamps;  // -DMVPTR_TIMELINE_BUILD only (MVPTR_GEMM_STAMPS)
};

// Up to MVPTR_TN_MAX_GROUP independent problems served by one launch: workgroup ids
// [base[i], base[i+1]) belong to problem i.  The workgroups of a problem that finish early hand
// their CU to the next problem's, so only the last problem's atomic write-out is an exposed tail.
struct GemmTnGroup {
  GemmTnArgs prob[MVPTR_TN_MAX_GROUP];
  int base[MVPTR_TN_MAX_GROUP + 1];
  int count;
  int splits;
  // "Q" kernel, slab write-out: workgroup g stores its 256x256 f32 partial tile as slab g (256 KiB, in register order)
  // of this caller-provided buffer and tn_reduce_kernel adds a tile's slabs in split order; NULL = f32 atomics
  float* slab;
  // device-side row count (sync-free joint pass): the rows actually present (<= prob[*].M); every split then covers
  // ceil(rows / splits) rows rounded up to `tm_round` instead of the host's rows_per_split.  NULL: prob[*].M rows.
  const int* rows_dev;
  int tm_round;
};

// rows of this launch and rows per M-split: the host's plan, or — with a device-side count — the same number of splits
// over the rows that are really there
__device__ __forceinline__ void tn_split_rows(const GemmTnGroup& grp, const GemmTnArgs& p, int& Mv, int& rps) {
  Mv = p.M;
  rps = p.rows_per_split;
  if (grp.rows_dev != nullptr) {
    Mv = min(p.M, __builtin_amdgcn_readfirstlane(*grp.rows_dev));
    rps = ((Mv + grp.splits - 1) / grp.splits + grp.tm_round - 1) / grp.tm_round * grp.tm_round;
    rps = max(rps, grp.tm_round);
  }
}

__device__ __forceinline__ int swz256(int row) { return ((row & 3) << 2) | ((row >> 2) & 3); }

__device__ __forceinline__ bf16x8 tr_frag(const char* tile, uint32_t off_lo, uint32_t off_hi) {
  // speed ablations only (wrong values; diagnostic builds, tools/exp_tn.py):
  //   MVPTR_TN_EXP 1: one 16-byte read per fragment instead of two transposed 8-byte reads
  //   2: no MFMA   3: no LDS reads   4: no LDS-DMA loads   5: no atomic write-out
#if defined(MVPTR_TN_EXP) && MVPTR_TN_EXP == 1
  (void)off_hi;
  return *reinterpret_cast<const bf16x8*>(tile + (off_lo & ~15u));
#endif
#if defined(MVPTR_TN_EXP) && MVPTR_TN_EXP == 3
  bf16x8 c;
  for (int e = 0; e < 8; ++e) c[e] = f2bf((float)((off_lo + off_hi + e) & 7));
  asm volatile("" : "+v"(c));
  return c;
#endif
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
template <int TM_, int KSUB, int STAGES>
__global__ __launch_bounds__(512, ((TM_ == 32 && KSUB == 1) ? 4 : 2)) void gemm_tn_kernel(GemmTnGroup grp) {
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
  // 1-D grid over (split, tile).  Workgroups are dealt to the 8 XCDs round-robin by their linear
  // id; the bijective remap hands each XCD a CONTIGUOUS run of (split, tile) pairs, split-major,
  // so the workgroups that share an XCD's L2 sweep the same token rows at the same time: the dY
  // panel is shared by the tiles_k tiles of a row of tiles, the X panel by the tiles_n of a column.
  const int gidx = xcd_remap(blockIdx.x, gridDim.x);
  int pi = 0;
#pragma unroll
  for (int i = 1; i < MVPTR_TN_MAX_GROUP; ++i)
    if (i < grp.count && gidx >= grp.base[i]) pi = i;
  const GemmTnArgs& p = grp.prob[pi];
  const int nt = p.tiles_n * p.tiles_k;
  const int idx = gidx - grp.base[pi];
  const int split = idx / nt;
  const int t = idx - split * nt;
  const int tn = t / p.tiles_k;
  const int tk = t - tn * p.tiles_k;
  const int n0 = tn * TN_, k0 = tk * TKW;
  int Mv, rps;
  tn_split_rows(grp, p, Mv, rps);
  const int m_begin = split * rps;
  const int m_end = min(Mv, m_begin + rps);
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
#if defined(MVPTR_TN_EXP) && MVPTR_TN_EXP == 4
    return;
#endif
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
  // transposed-read offsets of sub-step 0; sub-step s adds 16 rows = s * 4096 bytes (the swizzle
  // only depends on row bits 0-3, which the 16-row step leaves alone)
  uint32_t ta[NBLK][2], tb[2][2];
#pragma unroll
  for (int hl = 0; hl < 2; ++hl) {
    const int row = 8 * h + 4 * hl + q;
#pragma unroll
    for (int b = 0; b < NBLK; ++b) {
      const int cha = (ncol_w % 128) / 8 + b * 4 + 2 * cb + (pp >> 1);
      ta[b][hl] = a_sub + row * 256 + ((cha ^ swz256(row)) << 4) + 8 * (pp & 1);
    }
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      const int chb = (kcol_w % 128) / 8 + b * 4 + 2 * cb + (pp >> 1);
      tb[b][hl] = b_sub + row * 256 + ((chb ^ swz256(row)) << 4) + 8 * (pp & 1);
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
  constexpr int LPS = NA + NB, AHEAD = STAGES - 1;
#pragma unroll
  for (int i = 0; i < AHEAD; ++i)
    if (i < nsteps) stage(i, i * TM_);
  int buf = 0;
  for (int st = 0; st < nsteps; ++st) {
    // step st has landed once only the loads of the (up to AHEAD-1) younger steps remain
    if (AHEAD >= 2 && st + 1 < nsteps)
      asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"((AHEAD - 1) * LPS) : "memory");
    else
      asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
    if (st + AHEAD < nsteps) {
      int nb = buf + AHEAD;
      if (nb >= STAGES) nb -= STAGES;
      stage(nb, (st + AHEAD) * TM_);
    }
    const char* la = lds + buf * STAGE_B;
    const char* lb = la + 2 * SUB_B;
    // software pipeline over the 16-row sub-steps: the transposed reads of sub-step s+1 are issued
    // before the MFMAs of sub-step s (two 8-byte reads per fragment: twice the LDS instructions of a
    // row-major operand, and a wave can only keep 15 of them in flight)
    bf16x8 af[2][NBLK], bfr[2][2];
#pragma unroll
    for (int b = 0; b < NBLK; ++b) af[0][b] = tr_frag(la, ta[b][0], ta[b][1]);
#pragma unroll
    for (int b = 0; b < 2; ++b) bfr[0][b] = tr_frag(lb, tb[b][0], tb[b][1]);
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      const int cur = s & 1, nxt = cur ^ 1;
      if (s + 1 < NS) {
#pragma unroll
        for (int b = 0; b < NBLK; ++b) af[nxt][b] = tr_frag(la + (s + 1) * 4096, ta[b][0], ta[b][1]);
#pragma unroll
        for (int b = 0; b < 2; ++b) bfr[nxt][b] = tr_frag(lb + (s + 1) * 4096, tb[b][0], tb[b][1]);
      }
#pragma unroll
      for (int nb = 0; nb < NBLK; ++nb)
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#if defined(MVPTR_TN_EXP) && MVPTR_TN_EXP == 2
          asm volatile("" ::"v"(af[cur][nb]), "v"(bfr[cur][kb]));
#else
          acc[nb][kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[cur][nb], bfr[cur][kb], acc[nb][kb], 0, 0, 0);
#endif
      if (do_bias) {  // 4 x v_dot2c_f32_bf16 against (1, 1) per fragment
        const bf16x2 ones = {f2bf(1.f), f2bf(1.f)};
#pragma unroll
        for (int nb = 0; nb < NBLK; ++nb)
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const bf16x2 pr = {af[cur][nb][2 * j], af[cur][nb][2 * j + 1]};
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
#if defined(MVPTR_TN_EXP) && MVPTR_TN_EXP == 5
        if (n < p.N && k < p.K && acc[nb][kb][r] == 12345.678f) p.dW[(int64_t)n * p.ldw + k] = 1.f;
#else
        if (n < p.N && k < p.K) atomicAdd(p.dW + (int64_t)n * p.ldw + k, acc[nb][kb][r]);
#endif
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
    unsigned long long* o = p.stamps + (size_t)blockIdx.x * 8;
    o[0] = tl_start;
    o[1] = tl_loop;
    o[2] = tl_end;
  }
#endif
}

// "Q" configuration: 256(n) x 256(k) output tile per 256-thread workgroup, FOUR waves as 2(n) x 2(k),
// each 128x128 = 4x4 v_mfma_f32_32x32x16_bf16 (256 accumulator registers, one wave per SIMD with the
// whole 512-register file).  Against the 64x64 wave tiles above this halves the transposed LDS
// reads per MFMA (16 ds_read_b64_tr_b16 per 16 MFMAs) and against the 256x128 tile it needs a third
// fewer L2->LDS bytes per FLOP — the two resources the ablation builds showed to be co-limiting.
// 32 token rows per stage (32 KiB), STAGES-deep ring.  A lone wave per SIMD has no partner to cover
// its barrier / LDS-DMA issue / fragment-read latency, so the loop is rotated: the barrier that
// certifies stage st+1 sits between the two 16-row halves of stage st, and what follows it (issue of
// stage st+STAGES into the buffer just freed, fragment reads of the next stage's first half) runs
// under the 16 MFMAs of the second half, whose fragments are already in registers.
template <int STAGES, bool SLAB>
__global__ __launch_bounds__(256, 1) void gemm_tn_q_kernel(GemmTnGroup grp) {
  constexpr int TM_ = 32;
  constexpr int SUB_B = TM_ * 256;     // one 32 x 128 bf16 sub-tile (8 KiB)
  constexpr int STAGE_B = 4 * SUB_B;   // A: 2 sub-tiles (256 n), B: 2 sub-tiles (256 k)
  constexpr int NI = 4, LPS = 8;       // staging instructions per wave: 4 for A, 4 for B
  constexpr int TKW = 256;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int gidx = xcd_remap(blockIdx.x, gridDim.x);
  // problem lookup by uniform selects (a dynamically indexed kernel-argument array would be copied
  // to scratch and fetched on the vector path, which turns every buffer operation into a waterfall loop)
  GemmTnArgs p = grp.prob[0];
  int pbase = 0;
#pragma unroll
  for (int i = 1; i < MVPTR_TN_MAX_GROUP; ++i)
    if (i < grp.count && gidx >= grp.base[i]) {
      p = grp.prob[i];
      pbase = grp.base[i];
    }
  const int nt = p.tiles_n * p.tiles_k;
  const int idx = gidx - pbase;
  const int split = idx / nt;
  const int t = idx - split * nt;
  // tile order inside a split: the LONGER tile dimension outside, the shorter inside, so that the
  // contiguous run of tiles an XCD gets is a compact block of the (tn, tk) grid and re-reads the
  // fewest operand panels through its L2 (FFN2's 3 x 12 grid: 18 panel reads per split instead of 25)
  int tn, tk;
  if (p.tiles_n < p.tiles_k && !p.order_n_major) {
    tk = t / p.tiles_n;
    tn = t - tk * p.tiles_n;
  } else {
    tn = t / p.tiles_k;
    tk = t - tn * p.tiles_k;
  }
  const int n0 = tn * TN_, k0 = tk * TKW;
  int Mv, rps;
  tn_split_rows(grp, p, Mv, rps);
  const int m_begin = split * rps;
  const int m_end = min(Mv, m_begin + rps);
  const int rows = m_end - m_begin;
  if (rows <= 0) return;
  const int ncols = min(TN_, p.N - n0);
  const int kcols = min(TKW, p.K - k0);
  const int ncols8 = (int)min((int64_t)((ncols + 7) & ~7), p.lda - n0);
  const int kcols8 = (int)min((int64_t)((kcols + 7) & ~7), p.ldb - k0);
  const u32x4 rsA = make_rsrc_words(
      p.A + (int64_t)m_begin * p.lda + n0, (uint32_t)(((int64_t)(rows - 1) * p.lda + ncols8) * 2));
  const u32x4 rsB = make_rsrc_words(
      p.B + (int64_t)m_begin * p.ldb + k0, (uint32_t)(((int64_t)(rows - 1) * p.ldb + kcols8) * 2));
  const uint32_t lds0 = lds_addr(lds);

  // staging instruction i of this wave fills LDS KiB (i * 4 + wave) of the operand: sub-tile
  // (i * 4 + wave) / 8, 4-row group (i * 4 + wave) % 8
  uint32_t offA[NI], offB[NI];
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    const int j = i * 4 + wave;
    const int row = (j & 7) * 4 + (lane >> 4);
    const int ch = (lane & 15) ^ swz256(row);
    const int col = (j >> 3) * 128 + ch * 8;
    offA[i] = (col < ncols) ? (uint32_t)(row * p.lda * 2 + col * 2) : MVPTR_OOB;
    offB[i] = (col < kcols) ? (uint32_t)(row * p.ldb * 2 + col * 2) : MVPTR_OOB;
  }
  const uint32_t stepA = (uint32_t)(TM_ * p.lda * 2), stepB = (uint32_t)(TM_ * p.ldb * 2);
  // The LDS-DMA loads are inline asm: hipcc does not count them, so it neither drains them with a
  // vmcnt(0) in front of the next ds_read (it does for the builtin form) nor needs to know their
  // count; every wait for them is the hand-counted one in front of the barrier.
  // piece i (0..7) of a stage: A instructions 0..3, then B instructions 0..3
  auto stage_piece = [&](int buf, int st, int i) {
    const uint32_t la = lds0 + (uint32_t)(buf * STAGE_B + wave * 1024);
    if (i < NI) {
      const uint32_t va = (offA[i] == MVPTR_OOB) ? MVPTR_OOB : offA[i] + (uint32_t)st * stepA;
      lds_dma16(rsA, va, la + i * 4096);
    } else {
      const int j = i - NI;
      const uint32_t vb = (offB[j] == MVPTR_OOB) ? MVPTR_OOB : offB[j] + (uint32_t)st * stepB;
      lds_dma16(rsB, vb, la + 2 * SUB_B + j * 4096);
    }
  };
  auto stage = [&](int buf, int st) {
#pragma unroll
    for (int i = 0; i < 2 * NI; ++i) stage_piece(buf, st, i);
  };

  const int wn = wave >> 1, wk = wave & 1;
  const int g = lane >> 4, i16 = lane & 15;
  const int h = g >> 1, cb = g & 1;
  const int q = i16 >> 2, pp = i16 & 3;
  // transposed-read offsets of the first 16-row half; the second half adds 4096 bytes
  uint32_t ta[4][2], tb[4][2];
#pragma unroll
  for (int hl = 0; hl < 2; ++hl) {
    const int row = 8 * h + 4 * hl + q;
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      const int ch = b * 4 + 2 * cb + (pp >> 1);
      const uint32_t o = row * 256 + ((ch ^ swz256(row)) << 4) + 8 * (pp & 1);
      ta[b][hl] = (uint32_t)wn * SUB_B + o;
      tb[b][hl] = (uint32_t)(2 + wk) * SUB_B + o;
    }
  }

  f32x16 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  const bool do_bias = (p.colsum != nullptr) && (tk == 0) && (wk == 0);
  float bsum[4] = {0.f, 0.f, 0.f, 0.f};

  bf16x8 fa0[4], fb0[4], fa1[4], fb1[4];
  auto read_frags = [&](const char* base, bf16x8(&fa)[4], bf16x8(&fb)[4]) {
#pragma unroll
    for (int b = 0; b < 4; ++b) fa[b] = tr_frag(base, ta[b][0], ta[b][1]);
#pragma unroll
    for (int b = 0; b < 4; ++b) fb[b] = tr_frag(base, tb[b][0], tb[b][1]);
  };
  // fragment reads in the order B0 B1 | B2 B3 | A0 A1 | A2 A3 (one pair per quarter): a quarter's
  // MFMAs use A-fragment j with all four B fragments, so everything the next half's first quarter
  // needs has been requested at least a quarter (4 MFMAs) earlier
  auto read_pair = [&](const char* base, int j, bf16x8(&fa)[4], bf16x8(&fb)[4]) {
    if (j < 2) {
      fb[2 * j] = tr_frag(base, tb[2 * j][0], tb[2 * j][1]);
      fb[2 * j + 1] = tr_frag(base, tb[2 * j + 1][0], tb[2 * j + 1][1]);
    } else {
      fa[2 * j - 4] = tr_frag(base, ta[2 * j - 4][0], ta[2 * j - 4][1]);
      fa[2 * j - 3] = tr_frag(base, ta[2 * j - 3][0], ta[2 * j - 3][1]);
    }
  };
  // BIAS is a compile-time tag: a run-time branch inside the MFMA loop made hipcc spill the loop's
  // LDS offsets to scratch (whose reloads count in vmcnt beside the LDS-DMA loads)
  auto mma_row = [&](int nb, const bf16x8(&fa)[4], const bf16x8(&fb)[4], auto bias_tag) {
#pragma unroll
    for (int kb = 0; kb < 4; ++kb)
      acc[nb][kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[nb], fb[kb], acc[nb][kb], 0, 0, 0);
    if constexpr (decltype(bias_tag)::value) {
      const bf16x2 ones = {f2bf(1.f), f2bf(1.f)};
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const bf16x2 pr = {fa[nb][2 * j], fa[nb][2 * j + 1]};
        bsum[nb] = __builtin_amdgcn_fdot2_f32_bf16(pr, ones, bsum[nb], false);
      }
    }
  };

  const int nsteps = (rows + TM_ - 1) / TM_;
#pragma unroll
  for (int i = 0; i < STAGES; ++i)
    if (i < nsteps) stage(i, i);
  // stage 0 has landed once only the younger issued stages remain outstanding
  {
    const int younger = min(STAGES, nsteps) - 1;
    if (younger >= 3) asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(3 * LPS) : "memory");
    else if (younger == 2) asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(2 * LPS) : "memory");
    else if (younger == 1) asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(LPS) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
  }
  read_frags(lds, fa0, fb0);
  auto main_loop = [&](auto bias_tag) {
    int buf = 0;
    // one 32-row stage = two 16-row halves of four quarters (4 MFMAs each).  First half: MFMAs on
    // F0 beside the reads of F1 (second 16 rows of this stage).  Then the barrier.  Second half:
    // MFMAs on F1 beside the LDS-DMA issue of stage st+STAGES (into the buffer just freed) and the
    // reads of the next stage's F0.  sched_barrier(0) pins the quarters: hipcc otherwise sinks the
    // reads below the MFMAs and the wave then waits out a full LDS latency with an idle matrix pipe.
    // STEADY: the stage STAGES ahead exists and STAGES-2 younger stages stay in flight.
    auto step = [&](int st, auto steady_tag) {
      constexpr bool STEADY = decltype(steady_tag)::value;
      const char* cur = lds + buf * STAGE_B;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        read_pair(cur + 4096, j, fa1, fb1);
        mma_row(j, fa0, fb0, bias_tag);
        __builtin_amdgcn_sched_barrier(0);
      }
      const int nbuf = (buf + 1 == STAGES) ? 0 : buf + 1;
      const bool more = STEADY || st + 1 < nsteps;
      if (more) {
        // this wave's reads of stage st are complete (lgkmcnt(0), as a builtin so that hipcc knows
        // F1 has landed and puts no wait between the barrier and its MFMAs); after the barrier
        // every wave's are, and the buffer can be refilled
        __builtin_amdgcn_s_waitcnt(0xC07F);
        if constexpr (STEADY) {
          asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"((STAGES - 2) * LPS) : "memory");
        } else {
          const int younger = min(STAGES - 2, nsteps - 2 - st);
          if (younger >= 2) asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(2 * LPS) : "memory");
          else if (younger == 1) asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(LPS) : "memory");
          else asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
        }
      }
      const char* nxt = lds + nbuf * STAGE_B;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if constexpr (STEADY) {
          stage_piece(buf, st + STAGES, 2 * j);
          stage_piece(buf, st + STAGES, 2 * j + 1);
        }
        if (more) read_pair(nxt, j, fa0, fb0);
        mma_row(j, fa1, fb1, bias_tag);
        __builtin_amdgcn_sched_barrier(0);
      }
      buf = nbuf;
    };
    int st = 0;
    for (; st + STAGES < nsteps; ++st) step(st, std::true_type{});
    for (; st < nsteps; ++st) step(st, std::false_type{});
  };
  if (do_bias) main_loop(std::true_type{});
  else main_loop(std::false_type{});

  // write-out: f32 atomics, one accumulator register = two 128-B row segments per wave instruction.
  // The lane-derived addresses are formed from an opaque copy of the lane id so that they are
  // computed here and not hoisted above the main loop (256 live addresses would be spilled).
  int lane_e = lane;
  asm volatile("" : "+v"(lane_e));
  const int l31 = lane_e & 31, hh = lane_e >> 5;
  if constexpr (SLAB) {
    // slab mode (few-row launches, where the atomics of the M-splits are a quarter to a third of the launch): the
    // partial tile goes out as plain 16-byte stores, 1 KiB contiguous per wave instruction (float index
    // ((((wave*4 + nb)*4 + kb)*4 + i)*64 + lane)*4 + j holds accumulator register r = 4 i + j); the sum over the
    // splits is taken by tn_reduce_kernel in split order: bitwise reproducible
    float* sl = grp.slab + (int64_t)gidx * 65536 + wave * 16384 + lane_e * 4;
#pragma unroll
    for (int nb = 0; nb < 4; ++nb)
#pragma unroll
      for (int kb = 0; kb < 4; ++kb)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const f32x4 v = {acc[nb][kb][4 * i], acc[nb][kb][4 * i + 1], acc[nb][kb][4 * i + 2], acc[nb][kb][4 * i + 3]};
          *reinterpret_cast<f32x4*>(sl + ((nb * 4 + kb) * 4 + i) * 256) = v;
        }
    if (do_bias) {
#pragma unroll
      for (int nb = 0; nb < 4; ++nb) {
        const float tot = bsum[nb] + __shfl_xor(bsum[nb], 32);
        const int n = n0 + wn * 128 + nb * 32 + l31;
        if (hh == 0 && n < p.N) atomicAdd(p.colsum + n, tot);
      }
    }
    return;
  }
  const int nw = n0 + wn * 128 + 4 * hh, kw = k0 + wk * 128 + l31;
  const bool full = (n0 + TN_ <= p.N) && (k0 + TKW <= p.K);
  float* wbase = p.dW + (int64_t)nw * p.ldw + kw;
  if (full) {
#pragma unroll
    for (int nb = 0; nb < 4; ++nb)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        float* rowp = wbase + (int64_t)(nb * 32 + (r & 3) + 8 * (r >> 2)) * p.ldw;
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) atomicAdd(rowp + kb * 32, acc[nb][kb][r]);
      }
  } else {
#pragma unroll
    for (int nb = 0; nb < 4; ++nb)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int nn = nb * 32 + (r & 3) + 8 * (r >> 2);
        float* rowp = wbase + (int64_t)nn * p.ldw;
#pragma unroll
        for (int kb = 0; kb < 4; ++kb)
          if (nw + nn < p.N && kw + kb * 32 < p.K) atomicAdd(rowp + kb * 32, acc[nb][kb][r]);
      }
  }
  if (do_bias) {
#pragma unroll
    for (int nb = 0; nb < 4; ++nb) {
      const float tot = bsum[nb] + __shfl_xor(bsum[nb], 32);
      const int n = n0 + wn * 128 + nb * 32 + l31;
      if (hh == 0 && n < p.N) atomicAdd(p.colsum + n, tot);
    }
  }
}


// ---------------------------------------------------------------------------------------------------------------
// Round 5: every weight gradient of an encoder STACK in one balanced launch ("SK").
//
// The per-layer launches above cut M into splits so that (tiles x splits) fills the chip; every split writes its own
// 256 x 256 f32 partial tile (atomics, or slabs + a reduce kernel): at the packed row counts of the text / visual stacks that
// write-out was 66 MB per launch against 35 MB of result and a quarter of the launch time.  Here the caller keeps the
// operands (dY, X) of ALL layers alive until the stack's backward pass is done and hands over the whole list (up to
// MVPTR_TN_STACK_MAX problems that share M): T = sum of the problems' 256 x 256 output tiles, G workgroups (one per CU, each
// keeps its CU).  Workgroup g sweeps ALL token rows of tile r * G + g in round r = 0 .. T / G - 1 (one contributor per tile:
// nothing to combine), and the T mod G tiles that are left over are cut into G equal runs of 32-row steps (a run may span a
// tile boundary), so every workgroup does the same number of steps: no wave quantisation, no idle tail, at most two
// partial tiles per workgroup.  The next segment's first stages are requested before the finished tile is written out
// (f32 atomics; the accumulators stay live meanwhile), so the write-out runs under the operand fill.  The workgroups of
// an XCD take 32 consecutive tiles of a round — a compact block of one problem's (tn, tk) grid, longer dimension outside —
// and start them at row 0 together: the dY / X panels they share come once from HBM and otherwise from the XCD's L2.
// Main loop = gemm_tn_q_kernel's (4 waves of 128 x 128, rotated 32-row stages, hand-counted LDS-DMA ring).
struct TnSkProb {      // 64 bytes
  const __bf16* A;
  const __bf16* B;
  float* dW;
  float* colsum;
  int lda, ldb, ldw;   // elements
  int N, K;
  int tiles_n, tiles_k;
  int tile_base;       // first linear tile index of this problem
};
struct TnSkArgs {
  int count, tiles_total, M, flat_rem;
  const int* rows_dev;
  TnSkProb prob[MVPTR_TN_STACK_MAX];
};
static_assert(sizeof(TnSkProb) == 64, "TnSkProb layout");
static_assert(sizeof(TnSkArgs) <= 4096, "kernel arguments must fit the kernarg segment");

template <int STAGES>
__global__ __launch_bounds__(256, 1) void gemm_tn_sk_kernel(TnSkArgs args) {
  constexpr int TM_ = 32;
  constexpr int SUB_B = TM_ * 256;
  constexpr int STAGE_B = 4 * SUB_B;
  constexpr int NI = 4, LPS = 8;
  constexpr int TKW = 256;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int G = gridDim.x;
  const int g = xcd_remap(blockIdx.x, G);
  const int Mv = rows_clamped(args.M, args.rows_dev);
  const int spt = (Mv + TM_ - 1) / TM_;     // 32-row steps of one tile's sweep
  if (spt <= 0) return;
  const int T = args.tiles_total;
  const int R = T / G, rem = T - R * G;
  // the problem table is read from the kernel-argument segment with scalar loads (a dynamically indexed by-value array
  // would be copied to scratch and fetched on the vector path: every buffer descriptor would then need a waterfall loop)
  typedef const __attribute__((address_space(4))) char* kchar_p;
  typedef const __attribute__((address_space(4))) TnSkProb* kprob_p;
  const kprob_p tab = (kprob_p)((kchar_p)__builtin_amdgcn_kernarg_segment_ptr() + offsetof(TnSkArgs, prob));
  const int count = args.count;
  const uint32_t lds0 = lds_addr(lds);

  const int wn = wave >> 1, wk = wave & 1;
  const int gq = lane >> 4, i16 = lane & 15;
  const int h = gq >> 1, cb = gq & 1;
  const int q = i16 >> 2, pp = i16 & 3;
  uint32_t ta[4][2], tb[4][2];
#pragma unroll
  for (int hl = 0; hl < 2; ++hl) {
    const int row = 8 * h + 4 * hl + q;
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      const int ch = b * 4 + 2 * cb + (pp >> 1);
      const uint32_t o = row * 256 + ((ch ^ swz256(row)) << 4) + 8 * (pp & 1);
      ta[b][hl] = (uint32_t)wn * SUB_B + o;
      tb[b][hl] = (uint32_t)(2 + wk) * SUB_B + o;
    }
  }

  f32x16 acc[4][4];
  float bsum[4] = {0.f, 0.f, 0.f, 0.f};
  bf16x8 fa0[4], fb0[4], fa1[4], fb1[4];
  auto read_frags = [&](const char* base, bf16x8(&fa)[4], bf16x8(&fb)[4]) {
#pragma unroll
    for (int b = 0; b < 4; ++b) fa[b] = tr_frag(base, ta[b][0], ta[b][1]);
#pragma unroll
    for (int b = 0; b < 4; ++b) fb[b] = tr_frag(base, tb[b][0], tb[b][1]);
  };
  auto read_pair = [&](const char* base, int j, bf16x8(&fa)[4], bf16x8(&fb)[4]) {
    if (j < 2) {
      fb[2 * j] = tr_frag(base, tb[2 * j][0], tb[2 * j][1]);
      fb[2 * j + 1] = tr_frag(base, tb[2 * j + 1][0], tb[2 * j + 1][1]);
    } else {
      fa[2 * j - 4] = tr_frag(base, ta[2 * j - 4][0], ta[2 * j - 4][1]);
      fa[2 * j - 3] = tr_frag(base, ta[2 * j - 3][0], ta[2 * j - 3][1]);
    }
  };
  auto mma_row = [&](int nb, const bf16x8(&fa)[4], const bf16x8(&fb)[4]) {
#pragma unroll
    for (int kb = 0; kb < 4; ++kb)
      acc[nb][kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[nb], fb[kb], acc[nb][kb], 0, 0, 0);
  };
  // column sums of the 16 rows x 128 dY columns in fa: 4 x v_dot2c_f32_bf16 against (1, 1) per fragment.  A real (scalar) branch
  // around it — the empty asm keeps hipcc from turning it into always-computed dot products + selects
  auto bias_rows = [&](const bf16x8(&fa)[4], bool duty) {
    if (duty) {
      asm volatile("" ::: "memory");
      const bf16x2 ones = {f2bf(1.f), f2bf(1.f)};
#pragma unroll
      for (int nb = 0; nb < 4; ++nb)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const bf16x2 pr = {fa[nb][2 * j], fa[nb][2 * j + 1]};
          bsum[nb] = __builtin_amdgcn_fdot2_f32_bf16(pr, ones, bsum[nb], false);
        }
    }
  };

  // tile t -> its problem (scalar scan of the table) and coordinates
  auto decode = [&](int t, TnSkProb& p, int& tn, int& tk) {
    int pi = 0;
#pragma clang loop vectorize(disable) unroll(disable)
    for (int i = 1; i < count; ++i)
      if (t >= tab[i].tile_base) pi = i;
    pi = __builtin_amdgcn_readfirstlane(pi);
    p.A = tab[pi].A;
    p.B = tab[pi].B;
    p.dW = tab[pi].dW;
    p.colsum = tab[pi].colsum;
    p.lda = tab[pi].lda;
    p.ldb = tab[pi].ldb;
    p.ldw = tab[pi].ldw;
    p.N = tab[pi].N;
    p.K = tab[pi].K;
    p.tiles_n = tab[pi].tiles_n;
    p.tiles_k = tab[pi].tiles_k;
    p.tile_base = tab[pi].tile_base;
    const int tl = t - p.tile_base;
    if (p.tiles_n < p.tiles_k) {
      tk = tl / p.tiles_n;
      tn = tl - tk * p.tiles_n;
    } else {
      tn = tl / p.tiles_k;
      tk = tl - tn * p.tiles_k;
    }
  };
  // The finished segment whose accumulators still wait for their write-out (it goes out behind the next segment's first
  // loads): only its tile index stays live across the loops, everything else is derived again from the table.
  int pend_t = -1;
  auto write_out = [&]() {
    TnSkProb p;
    int tn, tk;
    decode(pend_t, p, tn, tk);
    // f32 buffer atomics: ONE per-lane offset register (+ its column-masked copies), the row of every instruction is a
    // scalar offset — no per-row address registers (the pointer form of gemm_tn_q_kernel needs 250 VGPRs for them).  One
    // accumulator register = two 128-byte row segments per wave instruction.  Rows past N fall outside the descriptor,
    // columns past K are masked per lane.
    int lane_e = lane;
    asm volatile("" : "+v"(lane_e));
    const int l31 = lane_e & 31, hh = lane_e >> 5;
    const int nrow0 = tn * TN_ + wn * 128, kcol0 = tk * TKW + wk * 128;
    const int nleft = p.N - nrow0, kleft = p.K - kcol0;
    const uint32_t bytes = (nleft > 0 && kleft > 0) ? (uint32_t)(((int64_t)(nleft - 1) * p.ldw + min(kleft, 128)) * 4) : 0u;
    const __amdgpu_buffer_rsrc_t rs = make_rsrc_uniform(p.dW + (int64_t)nrow0 * p.ldw + kcol0, bytes);
    const uint32_t voff0 = (uint32_t)((4 * hh * p.ldw + l31) * 4);
    uint32_t voff[4];
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) voff[kb] = (kb * 32 + l31 < kleft) ? voff0 + (uint32_t)kb * 128u : MVPTR_OOB;
    const int row_b = __builtin_amdgcn_readfirstlane(p.ldw * 4);
#pragma unroll
    for (int nb = 0; nb < 4; ++nb)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int soff = (nb * 32 + (r & 3) + 8 * (r >> 2)) * row_b;
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) __builtin_amdgcn_raw_ptr_buffer_atomic_fadd_f32(acc[nb][kb][r], rs, voff[kb], soff, 0);
        __builtin_amdgcn_sched_barrier(0);   // four accumulator reads at a time (hipcc otherwise copies all 256 to VGPRs first and spills)
      }
    if (p.colsum != nullptr) {      // every wave of every tile carries a share of the column sums (see bias_ctr)
#pragma unroll
      for (int nb = 0; nb < 4; ++nb) {
        const float tot = bsum[nb] + __shfl_xor(bsum[nb], 32);
        const int n = nrow0 + nb * 32 + l31;
        if (hh == 0 && n < p.N) atomicAdd(p.colsum + n, tot);
      }
    }
  };

  // Remainder tiles (fewer than G): shared out per XCD so that the workgroups of one XCD sweep THE SAME ROWS of neighbouring tiles
  // at the same time (their operand panels then meet in that XCD's L2, as in the full rounds).  XCD x owns a contiguous run of
  // n_x remainder tiles and W of the G workgroups.  Phase by phase: S = the smallest divisor of W that is >= W / (tiles left);
  // the next W / S tiles are each cut into S equal row ranges; workgroup l takes tile l % (W / S), range l / (W / S).  n_x / W of
  // a tile's sweep per workgroup in total (17 / 32 = 1/2 + 1/32: two phases), every XCD within one S-th of a tile of the others.
  const int NX = G < 8 ? G : 8;
  const int xcd = (int)blockIdx.x % NX, lw = (int)blockIdx.x / NX;
  const int W = G / NX + (xcd < G % NX ? 1 : 0);
  int n_left = rem / NX + (xcd < rem % NX ? 1 : 0);
  int cursor = R * G + xcd * (rem / NX) + min(xcd, rem % NX);
  // diagnostic A/B (MVPTR_NT_EXP bit 11): the round-5a schedule — one run of steps [b0, b1) of the flat (tile, step) list per workgroup
  long long b0 = 0, b1 = 0;
  if (args.flat_rem && rem > 0) {
    const long long tot = (long long)rem * spt;
    b0 = tot * g / G;
    b1 = tot * (g + 1) / G;
  }
  int it = 0;
  // next segment of this workgroup: tile t, steps [s0, s1); false when there is none left
  auto next_segment = [&](int& t, int& s0, int& s1) -> bool {
    if (it < R) {
      t = it * G + g;
      s0 = 0;
      s1 = spt;
      ++it;
    } else if (args.flat_rem) {
      if (b0 >= b1) return false;
      const int ti = (int)(b0 / spt);
      s0 = (int)(b0 - (long long)ti * spt);
      s1 = (int)min((long long)spt, (long long)s0 + (b1 - b0));
      t = R * G + ti;
      b0 += s1 - s0;
    } else {
      for (;;) {
        if (n_left <= 0) return false;
        int S = (W + n_left - 1) / n_left;
        while (W % S) ++S;
        const int tp = W / S;
        const int ti = lw % tp, sp = lw / tp;
        t = cursor + ti;
        s0 = (int)((long long)spt * sp / S);
        s1 = (int)((long long)spt * (sp + 1) / S);
        cursor += tp;
        n_left -= tp;
        if (s1 > s0) break;
      }
    }
    t = __builtin_amdgcn_readfirstlane(t);
    s0 = __builtin_amdgcn_readfirstlane(s0);
    s1 = __builtin_amdgcn_readfirstlane(s1);
    return true;
  };
  // operand state of the CURRENT segment (set by begin_segment, used by the ring refills of its main loop)
  u32x4 rsA, rsB;
  uint32_t offA[NI], offB[NI];
  uint32_t stepA = 0, stepB = 0;
  int nsteps = 0;
  // Bias-gradient column sums of the dY operand (colsum != NULL): the 2 * tiles_k waves that hold the same dY columns (the two
  // k-halves of every tile along k) take the 32-row steps in turn — step s belongs to wave (tk, wk) = s mod (2 tiles_k) — so every
  // tile of the problem runs the same instruction mix.  Round 5a gave all of it to the (tk = 0, wk = 0) waves: 32 extra VALU
  // instructions per step made those tiles ~5 % slower than their neighbours, they fell out of the window in which an XCD's L2
  // still holds the shared dY panel, and the panel was fetched twice (FETCH_SIZE 2.23 x the operands for FFN1-type problems against
  // 1.38 x without column sums; the step's launches 3 030 -> 2 730 us at M = 37 748: profiles/r05_experiments.txt section 9).
  bool do_bias = false;
  int bias_mod = 1, bias_ctr = 0;
  auto stage_piece = [&](int buf, int st, int i) {
    const uint32_t la = lds0 + (uint32_t)(buf * STAGE_B + wave * 1024);
    if (i < NI) {
      const uint32_t va = (offA[i] == MVPTR_OOB) ? MVPTR_OOB : offA[i] + (uint32_t)st * stepA;
      lds_dma16(rsA, va, la + i * 4096);
    } else {
      const int j = i - NI;
      const uint32_t vb = (offB[j] == MVPTR_OOB) ? MVPTR_OOB : offB[j] + (uint32_t)st * stepB;
      lds_dma16(rsB, vb, la + 2 * SUB_B + j * 4096);
    }
  };
  // descriptors and per-lane offsets of segment (t, s0, s1), then its first STAGES stages are requested (every wave has
  // passed the barrier: the ring is free)
  auto begin_segment = [&](int t, int s0, int s1) {
    TnSkProb p;
    int tn, tk;
    decode(t, p, tn, tk);
    const int n0 = tn * TN_, k0 = tk * TKW;
    const int m_begin = s0 * TM_;
    const int m_end = min(Mv, s1 * TM_);
    const int rows = m_end - m_begin;
    nsteps = s1 - s0;
    const int ncols = min(TN_, p.N - n0);
    const int kcols = min(TKW, p.K - k0);
    const int ncols8 = min((ncols + 7) & ~7, p.lda - n0);
    const int kcols8 = min((kcols + 7) & ~7, p.ldb - k0);
    rsA = make_rsrc_words(p.A + (int64_t)m_begin * p.lda + n0, (uint32_t)(((int64_t)(rows - 1) * p.lda + ncols8) * 2));
    rsB = make_rsrc_words(p.B + (int64_t)m_begin * p.ldb + k0, (uint32_t)(((int64_t)(rows - 1) * p.ldb + kcols8) * 2));
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int j = i * 4 + wave;
      const int row = (j & 7) * 4 + (lane >> 4);
      const int ch = (lane & 15) ^ swz256(row);
      const int col = (j >> 3) * 128 + ch * 8;
      offA[i] = (col < ncols) ? (uint32_t)(row * p.lda * 2 + col * 2) : MVPTR_OOB;
      offB[i] = (col < kcols) ? (uint32_t)(row * p.ldb * 2 + col * 2) : MVPTR_OOB;
    }
    stepA = (uint32_t)(TM_ * p.lda * 2);
    stepB = (uint32_t)(TM_ * p.ldb * 2);
    // every problem runs the duty pattern, also those without column sums (their sums are dropped): the tiles of ALL problems
    // then keep the same pace, so the workgroups of an XCD still start their next tiles together after a round that mixed
    // problem types (FETCH_SIZE of the step's mix 1.51 x -> 1.32 x the operands, the same as without any column sums; the
    // launch alone is 1-5 % slower for it, the step the same: profiles/r05_experiments.txt section 9)
    do_bias = true;
    bias_mod = 2 * p.tiles_k;
    bias_ctr = (2 * tk + wk - s0 % bias_mod + bias_mod) % bias_mod;      // steps until this wave's next turn
    // every wave is done with the previous segment's ring (its fragment reads were consumed by its last MFMAs)
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#pragma unroll
    for (int i = 0; i < STAGES; ++i)
      if (i < nsteps) {
#pragma unroll
        for (int j = 0; j < 2 * NI; ++j) stage_piece(i, i, j);
      }
  };

  int t, s0, s1;
  if (!next_segment(t, s0, s1)) return;
  begin_segment(t, s0, s1);
  for (;;) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) bsum[i] = 0.f;
    // everything issued so far (this segment's first stages; before them the previous tile's atomics) has completed
    asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
    read_frags(lds, fa0, fb0);
    auto main_loop = [&](auto bias_tag) {
      int buf = 0;
      auto step = [&](int st, auto steady_tag) {
        constexpr bool STEADY = decltype(steady_tag)::value;
        const char* cur = lds + buf * STAGE_B;
        bool duty = false;
        if constexpr (decltype(bias_tag)::value) {
          duty = (bias_ctr == 0);
          bias_ctr = duty ? bias_mod - 1 : bias_ctr - 1;
        }
        if constexpr (decltype(bias_tag)::value) bias_rows(fa0, duty);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          read_pair(cur + 4096, j, fa1, fb1);
          mma_row(j, fa0, fb0);
          __builtin_amdgcn_sched_barrier(0);
        }
        const int nbuf = (buf + 1 == STAGES) ? 0 : buf + 1;
        const bool more = STEADY || st + 1 < nsteps;
        if (more) {
          __builtin_amdgcn_s_waitcnt(0xC07F);
          if constexpr (STEADY) {
            asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"((STAGES - 2) * LPS) : "memory");
          } else {
            const int younger = min(STAGES - 2, nsteps - 2 - st);
            if (younger >= 2) asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(2 * LPS) : "memory");
            else if (younger == 1) asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(LPS) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
          }
        }
        const char* nxt = lds + nbuf * STAGE_B;
        if constexpr (decltype(bias_tag)::value) {
          if (!more) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // last step: fa1 was not waited for at a barrier
          bias_rows(fa1, duty);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          if constexpr (STEADY) {
            stage_piece(buf, st + STAGES, 2 * j);
            stage_piece(buf, st + STAGES, 2 * j + 1);
          }
          if (more) read_pair(nxt, j, fa0, fb0);
          mma_row(j, fa1, fb1);
          __builtin_amdgcn_sched_barrier(0);
        }
        buf = nbuf;
      };
      int st = 0;
      for (; st + STAGES < nsteps; ++st) step(st, std::true_type{});
      for (; st < nsteps; ++st) step(st, std::false_type{});
    };
    if (do_bias) main_loop(std::true_type{});
    else main_loop(std::false_type{});
    // the next segment's first stages are requested in front of this tile's write-out: the atomics run under the fill
    pend_t = t;
    const bool more_segments = next_segment(t, s0, s1);
    if (more_segments) begin_segment(t, s0, s1);
    write_out();
    if (!more_segments) break;
  }
}

// Sum of the M-splits' slabs of gemm_tn_q_kernel into dW (+=), splits in ascending order: the result does not depend
// on the order in which workgroups finished (bitwise reproducible, unlike the atomic write-out).  One thread per
// 16-byte slab position = four rows n..n+3 of one column k; a wave reads 1 KiB contiguous per split and updates
// 2 x 128-byte row segments per row.
__global__ __launch_bounds__(256) void tn_reduce_kernel(GemmTnGroup grp) {
  const int splits = grp.splits;
  const int tile_lin = blockIdx.x >> 6;
  GemmTnArgs p = grp.prob[0];
  int pbase = 0;
#pragma unroll
  for (int i = 1; i < MVPTR_TN_MAX_GROUP; ++i)
    if (i < grp.count && tile_lin * splits >= grp.base[i]) {
      p = grp.prob[i];
      pbase = grp.base[i];
    }
  const int nt = p.tiles_n * p.tiles_k;
  const int t = tile_lin - pbase / splits;
  int tn, tk;
  if (p.tiles_n < p.tiles_k && !p.order_n_major) {
    tk = t / p.tiles_n;
    tn = t - tk * p.tiles_n;
  } else {
    tn = t / p.tiles_k;
    tk = t - tn * p.tiles_k;
  }
  const int e4 = (blockIdx.x & 63) * 256 + threadIdx.x;   // 16-byte position inside the tile's slab
  const int lane = e4 & 63;
  const f32x4* src = reinterpret_cast<const f32x4*>(grp.slab) + ((int64_t)(pbase + t) * 16384 + e4);
  const int64_t sstride = (int64_t)nt * 16384;
  // rows_per_split covers M with `active` splits (the planner's count may leave trailing ones empty)
  int Mv, rps;
  tn_split_rows(grp, p, Mv, rps);
  const int active = min(splits, (Mv + rps - 1) / rps);
  f32x4 sum = {0.f, 0.f, 0.f, 0.f};
  int sidx = 0;
  for (; sidx + 4 <= active; sidx += 4) {
    const f32x4 a = src[(int64_t)sidx * sstride], b = src[(int64_t)(sidx + 1) * sstride];
    const f32x4 c = src[(int64_t)(sidx + 2) * sstride], d = src[(int64_t)(sidx + 3) * sstride];
    sum += a;
    sum += b;
    sum += c;
    sum += d;
  }
  for (; sidx < active; ++sidx) sum += src[(int64_t)sidx * sstride];
  // e4 = (((wave*4 + nb)*4 + kb)*4 + i)*64 + lane, four waves 2(n) x 2(k)
  const int i = (e4 >> 6) & 3, kb = (e4 >> 8) & 3, nb = (e4 >> 10) & 3, wave = e4 >> 12;
  const int n = tn * TN_ + (wave >> 1) * 128 + nb * 32 + 8 * i + 4 * (lane >> 5);
  const int k = tk * 256 + (wave & 1) * 128 + kb * 32 + (lane & 31);
  if (k < p.K) {
    float* o = p.dW + (int64_t)n * p.ldw + k;
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (n + j < p.N) o[(int64_t)j * p.ldw] += sum[j];
  }
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

template <int TM_, int KSUB, int STAGES>
int launch_tn(const GemmTnGroup& g, hipStream_t stream) {
  const int lds_b = STAGES * (2 + KSUB) * TM_ * 256;
  hipError_t e = hipFuncSetAttribute((const void*)gemm_tn_kernel<TM_, KSUB, STAGES>,
                                     hipFuncAttributeMaxDynamicSharedMemorySize, lds_b);
  if (e != hipSuccess) MVPTR_FAIL(MVPTR_HIP_ERROR, "gemm_tn: set LDS size: %s", hipGetErrorString(e));
  hipLaunchKernelGGL((gemm_tn_kernel<TM_, KSUB, STAGES>), dim3(g.base[g.count]), dim3(512), lds_b, stream, g);
  MVPTR_CHECK_LAUNCH("gemm_tn");
  return MVPTR_OK;
}

template <int STAGES>
int launch_tn_q(GemmTnGroup& g, hipStream_t stream) {
  const int lds_b = STAGES * 4 * 32 * 256;
  const void* fn = g.slab ? (const void*)gemm_tn_q_kernel<STAGES, true> : (const void*)gemm_tn_q_kernel<STAGES, false>;
  hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, lds_b);
  if (e != hipSuccess) MVPTR_FAIL(MVPTR_HIP_ERROR, "gemm_tn: set LDS size: %s", hipGetErrorString(e));
  if (g.slab) {
    hipLaunchKernelGGL((gemm_tn_q_kernel<STAGES, true>), dim3(g.base[g.count]), dim3(256), lds_b, stream, g);
    MVPTR_CHECK_LAUNCH("gemm_tn");
    hipLaunchKernelGGL(tn_reduce_kernel, dim3((g.base[g.count] / g.splits) * 64), dim3(256), 0, stream, g);
    MVPTR_CHECK_LAUNCH("gemm_tn reduce");
    return MVPTR_OK;
  }
  hipLaunchKernelGGL((gemm_tn_q_kernel<STAGES, false>), dim3(g.base[g.count]), dim3(256), lds_b, stream, g);
  MVPTR_CHECK_LAUNCH("gemm_tn");
  return MVPTR_OK;
}

struct TnPlan {
  int tm, ksub, splits;
  double cost;
};

// Estimated time of one configuration for `tiles` output tiles of one launch (all problems of a
// group share M, so they share the split count): whole rounds of workgroups x steps per split,
// plus the f32 atomics of every M-split (~1.3 TB/s chip-wide, partly overlapped).
TnPlan plan_tn(int M, int64_t out_elems, int tiles, int tm, int ksub, bool quad = false, bool slab = false) {
  const int slots = (tm == 32 && ksub == 1) ? 512 : 256;
  // microseconds per 64 token rows of one workgroup (measured on MI355X, round 1; "Q": round 2)
  const double t64 = quad ? 1.5 : (ksub == 2) ? (tm == 64 ? 1.75 : 3.1) : (tm == 64 ? 1.7 : 2.7);
  TnPlan best{tm, ksub, 1, 1e30};
  const int max_splits = (M + 255) / 256;
  for (int sp = 1; sp <= 64 && sp <= max_splits; ++sp) {
    const double rounds = (double)((tiles * sp + slots - 1) / slots);
    const double steps = (double)((M + sp * 64 - 1) / (sp * 64));
    const double out_b = (double)out_elems * 4.0;
    // write-out: f32 atomics (~1.3 TB/s chip-wide, partly overlapped), or plain slab stores (~6.5 TB/s) + the reduce
    // kernel's reads of every slab (mostly L2 / Infinity-Cache hits) + its launch
    const double wr = (slab && sp >= 2) ? sp * out_b / 6.5e6 + (sp + 2) * out_b / 5.0e6 + 4.0 : sp * out_b / 1.3e6 * 0.7;
    const double cost = rounds * steps * t64 + wr;
    if (cost < best.cost) {
      best.cost = cost;
      best.splits = sp;
    }
  }
  return best;
}

int check_problem(const mvptr_tn_problem& q) {
  if (q.M <= 0 || q.N <= 0 || q.K <= 0) MVPTR_FAIL(MVPTR_BAD_SHAPE, "gemm_tn: M,N,K must be > 0");
  if ((q.lda & 7) || (q.ldb & 7)) MVPTR_FAIL(MVPTR_BAD_ALIGN, "gemm_tn: lda, ldb must be multiples of 8");
  if (q.lda < q.N || q.ldb < q.K) MVPTR_FAIL(MVPTR_BAD_SHAPE, "gemm_tn: lda/ldb smaller than N/K");
  if (((uintptr_t)q.A & 15) || ((uintptr_t)q.B & 15)) MVPTR_FAIL(MVPTR_BAD_ALIGN, "gemm_tn: A and B must be 16-byte aligned");
  if (!q.A || !q.B || !q.dW) MVPTR_FAIL(MVPTR_BAD_ARG, "gemm_tn: NULL argument");
  return MVPTR_OK;
}

// one launch for `count` problems with the same M
// Slab write-out of the "Q" kernel is used for launches of at most this many token rows (measured, round 2 tables in
// profiles/r02_experiments.txt: attention pair 91 -> 68 us and FFN pair 124 -> 111 us at M = 10 917, 118 -> 97 at 19 200;
// at 37 748 the FFN pair loses 3 %): where the atomics of the M-splits are a quarter to a third of the launch.
constexpr int TN_SLAB_MAX_ROWS = 24000;

// one launch for `count` problems with the same M.  ws / ws_bytes: caller's slab workspace (may be NULL); need != NULL:
// plan only — *need = slab bytes this group would use (0: atomics), nothing is launched
int run_group(const mvptr_tn_problem* probs, int count, hipStream_t stream, float* ws, int64_t ws_bytes, int64_t* need,
              bool allow_slab = true, const int* rows_dev = nullptr, int M_plan = 0) {
  const int Mb = probs[0].M;                                              // rows the buffers hold (bound)
  const int M = (rows_dev != nullptr && M_plan > 0 && M_plan < Mb) ? M_plan : Mb;   // rows the launch is planned for
  // four configurations (see gemm_tn_kernel); MVPTR_GEMM_TN = "32" | "64" | "k2" | "K" forces one
  // (tuning knob).  The 256x256 tiles measured no faster than 256x128 at two workgroups per CU on
  // this model's shapes, so only the 256x128 tiles compete by default.
  const int cfg_tm[5] = {32, 64, 32, 64, 32}, cfg_ks[5] = {1, 1, 2, 2, 2};
  TnPlan plans[5];
  bool slab_plan = false;
  for (int c = 0; c < 5; ++c) {
    int tiles = 0;
    int64_t out_elems = 0;
    for (int i = 0; i < count; ++i) {
      tiles += ((probs[i].N + TN_ - 1) / TN_) * ((probs[i].K + cfg_ks[c] * 128 - 1) / (cfg_ks[c] * 128));
      out_elems += (int64_t)probs[i].N * probs[i].K;
    }
    plans[c] = plan_tn(M, out_elems, tiles, cfg_tm[c], cfg_ks[c], c == 4);
    if (c == 4 && allow_slab && M >= 6000 && M <= TN_SLAB_MAX_ROWS && (ws != nullptr || need != nullptr) &&
        mvptr_knobs().gemm_tn[0] == 0 && (mvptr_knobs().nt_exp & (1 << 12)) == 0) {   // MVPTR_NT_EXP bit 12 (diagnostic build): atomics everywhere
      const TnPlan sl = plan_tn(M, out_elems, tiles, cfg_tm[c], cfg_ks[c], true, true);
      if (sl.splits >= 2) {
        plans[c] = sl;
        slab_plan = true;
      }
    }
  }
  // "Q" (256x256 tile, four waves of 128x128) wins from ~6 k token rows up (tools/sweep_tn.py,
  // tools/sweep_tn_group.py: 1.0-1.15 PF/s against 0.7-0.8 at M >= 19 k, +8 % at 11 k, -7 % at 3 k)
  int pick = (plans[1].cost < plans[0].cost) ? 1 : 0;
  if (M >= 6000) pick = 4;
  const char* env = mvptr_knobs().gemm_tn;
  if (env[0] != 0) pick = (env[0] == '6') ? 1 : (env[0] == 'k') ? 2 : (env[0] == 'K') ? 3 : (env[0] == 'q' || env[0] == 'Q') ? 4 : 0;
  TnPlan pl = plans[pick];
  if (mvptr_knobs().tn_splits > 0) pl.splits = min(mvptr_knobs().tn_splits, (M + 255) / 256);
  int rps = (M + pl.splits - 1) / pl.splits;
  rps = (rps + pl.tm - 1) / pl.tm * pl.tm;
  // keep each split's byte span below 2 GiB (buffer offsets are 32-bit)
  int64_t ldmax = 8;
  for (int i = 0; i < count; ++i) {
    if (probs[i].lda > ldmax) ldmax = probs[i].lda;
    if (probs[i].ldb > ldmax) ldmax = probs[i].ldb;
  }
  while ((int64_t)rps * ldmax * 2 >= (int64_t)0x7fffffff && rps > pl.tm) rps = (rps / 2 + pl.tm - 1) / pl.tm * pl.tm;
  int splits = (M + rps - 1) / rps;
  if (rows_dev != nullptr) {
    // the kernels spread the rows that are really there over `splits`; a split never spans more than ceil(Mb / splits)
    // rows (+ rounding), which must stay below 2 GiB of operand bytes
    while ((int64_t)((Mb + splits - 1) / splits + pl.tm) * ldmax * 2 >= (int64_t)0x7fffffff) ++splits;
  }
  GemmTnGroup g;
  g.count = count;
  g.splits = splits;
  g.rows_dev = rows_dev;
  g.tm_round = pl.tm;
  g.base[0] = 0;
  for (int i = 0; i < count; ++i) {
    GemmTnArgs& a = g.prob[i];
    a.A = (const __bf16*)probs[i].A;
    a.B = (const __bf16*)probs[i].B;
    a.lda = probs[i].lda;
    a.ldb = probs[i].ldb;
    a.M = Mb;
    a.N = probs[i].N;
    a.K = probs[i].K;
    a.dW = probs[i].dW;
    a.ldw = probs[i].ldw;
    a.colsum = probs[i].colsum;
    a.order_n_major = (mvptr_knobs().nt_exp & 32) ? 1 : 0;
    a.stamps = nullptr;
    a.tiles_n = (a.N + TN_ - 1) / TN_;
    a.tiles_k = (a.K + pl.ksub * 128 - 1) / (pl.ksub * 128);
    a.rows_per_split = rps;
    g.base[i + 1] = g.base[i] + a.tiles_n * a.tiles_k * splits;
  }
  for (int i = count; i < MVPTR_TN_MAX_GROUP; ++i) {
    g.prob[i] = g.prob[0];
    g.base[i + 1] = g.base[count];
  }
#ifdef MVPTR_TIMELINE_BUILD
  for (int i = 0; i < MVPTR_TN_MAX_GROUP; ++i) g.prob[i].stamps = (unsigned long long*)mvptr_knobs().stamps;
#endif
  g.slab = nullptr;
  const bool slab = slab_plan && pick == 4 && splits >= 2;
  const int64_t slab_bytes = slab ? (int64_t)g.base[count] * 65536 * (int64_t)sizeof(float) : 0;
  if (need != nullptr) {
    *need = slab_bytes;
    return MVPTR_OK;
  }
  if (slab) {
    if (ws == nullptr || ws_bytes < slab_bytes || ((uintptr_t)ws & 15))
      return run_group(probs, count, stream, nullptr, 0, nullptr, false, rows_dev, M_plan);   // workspace too small: atomics with their own split plan
    g.slab = ws;
  }
  if (pick == 4) return launch_tn_q<4>(g, stream);
  if (pl.ksub == 2 && pl.tm == 64) return launch_tn<64, 2, 2>(g, stream);
  if (pl.ksub == 2) return launch_tn<32, 2, 3>(g, stream);
  if (pl.tm == 64) return launch_tn<64, 1, 3>(g, stream);
  return launch_tn<32, 1, 3>(g, stream);
}

}  // namespace

namespace {
int tn_multi(const mvptr_tn_problem* probs, int count, void* ws, int64_t ws_bytes, int64_t* need, void* stream,
             const int* rows_dev = nullptr, int M_plan = 0) {
  if (!probs || count <= 0) MVPTR_FAIL(MVPTR_BAD_ARG, "gemm_tn_multi: no problems");
  for (int i = 0; i < count; ++i) {
    const int rc = check_problem(probs[i]);
    if (rc != MVPTR_OK) return rc;
  }
  // problems are grouped while they share M (one split plan per launch); MVPTR_TN_GROUP=0 issues
  // them one by one (A/B knob, diagnostic build)
  const int max_group = (mvptr_knobs().tn_group == 0) ? 1 : MVPTR_TN_MAX_GROUP;
  int64_t most = 0;
  int i = 0;
  while (i < count) {
    int j = i + 1;
    while (j < count && j - i < max_group && probs[j].M == probs[i].M) ++j;
    int64_t nb = 0;
    const int rc = run_group(probs + i, j - i, (hipStream_t)stream, (float*)ws, ws_bytes, need ? &nb : nullptr, true, rows_dev, M_plan);
    if (rc != MVPTR_OK) return rc;
    if (nb > most) most = nb;   // the groups of one call run one after the other on the stream: they share the buffer
    i = j;
  }
  if (need) *need = most;
  return MVPTR_OK;
}
}  // namespace

extern "C" int mvptr_gemm_tn_multi(const mvptr_tn_problem* probs, int count, void* stream) {
  return tn_multi(probs, count, nullptr, 0, nullptr, stream);
}

extern "C" int mvptr_gemm_tn_multi_ws(const mvptr_tn_problem* probs, int count, void* ws, int64_t ws_bytes, void* stream) {
  return tn_multi(probs, count, ws, ws_bytes, nullptr, stream);
}

int mvptr_gemm_tn_multi_rows(const mvptr_tn_problem* probs, int count, void* ws, int64_t ws_bytes, const int* rows_dev, int M_plan,
                             void* stream) {
  return tn_multi(probs, count, ws, ws_bytes, nullptr, stream, rows_dev, M_plan);
}

extern "C" int64_t mvptr_gemm_tn_ws_bytes(const mvptr_tn_problem* probs, int count) {
  int64_t need = 0;
  if (tn_multi(probs, count, nullptr, 0, &need, nullptr) != MVPTR_OK) return -1;
  return need;
}

// Every weight gradient of an encoder stack in one balanced launch (gemm_tn_sk_kernel).  All problems share M (and the
// device-side row count, if any); lists longer than MVPTR_TN_STACK_MAX go out as several launches.
extern "C" int mvptr_gemm_tn_stack(const mvptr_tn_problem* probs, int count, const int* rows_dev, int max_workgroups, void* stream) {
  if (!probs || count <= 0) MVPTR_FAIL(MVPTR_BAD_ARG, "gemm_tn_stack: no problems");
  for (int i = 0; i < count; ++i) {
    const int rc = check_problem(probs[i]);
    if (rc != MVPTR_OK) return rc;
    if (probs[i].M != probs[0].M) MVPTR_FAIL(MVPTR_BAD_SHAPE, "gemm_tn_stack: the problems of a stack share M (%d vs %d)", probs[i].M, probs[0].M);
    if (probs[i].lda >= (int64_t)0x7fffffff / 64 || probs[i].ldb >= (int64_t)0x7fffffff / 64 || probs[i].ldw >= (int64_t)0x7fffffff)
      MVPTR_FAIL(MVPTR_BAD_SHAPE, "gemm_tn_stack: leading dimension too large");
  }
  // a workgroup's sweep addresses its rows with 32-bit byte offsets from the segment's first row
  {
    int64_t ldmax = 8;
    for (int i = 0; i < count; ++i) {
      if (probs[i].lda > ldmax) ldmax = probs[i].lda;
      if (probs[i].ldb > ldmax) ldmax = probs[i].ldb;
    }
    if ((int64_t)probs[0].M * ldmax * 2 >= (int64_t)0x7fffffff)
      MVPTR_FAIL(MVPTR_BAD_SHAPE, "gemm_tn_stack: M x leading dimension exceeds 2 GiB of operand bytes (use mvptr_gemm_tn_multi)");
  }
  int ncu = 256;
  {
    int dev = 0, v = 0;
    if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) ncu = v;
  }
  int G = (max_workgroups > 0 && max_workgroups < ncu) ? max_workgroups : ncu;
  constexpr int STAGES = 4;
  const int lds_b = STAGES * 4 * 32 * 256;
  {   // per call: the attribute belongs to the current device's copy of the kernel
    hipError_t e = hipFuncSetAttribute((const void*)gemm_tn_sk_kernel<STAGES>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_b);
    if (e != hipSuccess) MVPTR_FAIL(MVPTR_HIP_ERROR, "gemm_tn_stack: set LDS size: %s", hipGetErrorString(e));
  }
  for (int first = 0; first < count; first += MVPTR_TN_STACK_MAX) {
    const int n = (count - first < MVPTR_TN_STACK_MAX) ? count - first : MVPTR_TN_STACK_MAX;
    TnSkArgs a;
    memset(&a, 0, sizeof(a));
    a.count = n;
    a.M = probs[0].M;
    a.rows_dev = rows_dev;
    int base = 0;
    for (int i = 0; i < n; ++i) {
      const mvptr_tn_problem& q = probs[first + i];
      TnSkProb& t = a.prob[i];
      t.A = (const __bf16*)q.A;
      t.B = (const __bf16*)q.B;
      t.dW = q.dW;
      t.colsum = q.colsum;
      t.lda = (int)q.lda;
      t.ldb = (int)q.ldb;
      t.ldw = (int)q.ldw;
      t.N = q.N;
      t.K = q.K;
      t.tiles_n = (q.N + TN_ - 1) / TN_;
      t.tiles_k = (q.K + 255) / 256;
      t.tile_base = base;
      base += t.tiles_n * t.tiles_k;
    }
    for (int i = n; i < MVPTR_TN_STACK_MAX; ++i) {
      a.prob[i] = a.prob[0];
      a.prob[i].tile_base = 0x7fffffff;
    }
    a.tiles_total = base;
    a.flat_rem = (mvptr_knobs().nt_exp & (1 << 11)) ? 1 : 0;      // diagnostic build only (the product's knobs are constants)
    hipLaunchKernelGGL((gemm_tn_sk_kernel<STAGES>), dim3(G), dim3(256), lds_b, (hipStream_t)stream, a);
    MVPTR_CHECK_LAUNCH("gemm_tn_stack");
  }
  return MVPTR_OK;
}

extern "C" int mvptr_gemm_tn(const void* A, int64_t lda, const void* B, int64_t ldb, int M, int N,
                             int K, float* dW, int64_t ldw, float* colsum, void* stream) {
  mvptr_tn_problem q;
  q.A = A;
  q.lda = lda;
  q.B = B;
  q.ldb = ldb;
  q.M = M;
  q.N = N;
  q.K = K;
  q.dW = dW;
  q.ldw = ldw;
  q.colsum = colsum;
  return mvptr_gemm_tn_multi(&q, 1, stream);
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
