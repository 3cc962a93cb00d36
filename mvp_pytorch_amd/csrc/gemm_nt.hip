// gemm_nt.hip — C[M,N] = A[M,K] * B[N,K]^T (bf16 in, f32 accumulate) with fused epilogues.
//
// Reference op sequences replaced: nn.Linear forward + the elementwise ops that follow it in
// transformers/pytorch_transformers/modeling_bert.py:348-352 (dense+dropout+residual),
// :394-397 (dense+gelu :142-148), :407-411, oscar/modeling/modeling_vlbert.py:71-73 (Q/K/V),
// and the data-gradient GEMMs autograd derives from them.
//
// CDNA4 design
//  * 256(M) x 128(N) x 64(K) tile per 512-thread workgroup: 8 waves as 4(M) x 2(N), each wave a
//    64x64 block = 4x4 v_mfma_f32_16x16x32_bf16 tiles (64 accumulator registers).
//  * operands go HBM/L2 -> LDS with buffer_load ... lds (16 B per lane, no VGPR round trip);
//    out-of-range rows and the K tail come back as zeros from the buffer bounds check.
//  * 3-stage LDS ring (3 x 48 KiB): two K-steps stay in flight across the barrier behind a
//    counted s_waitcnt vmcnt(6) + raw s_barrier (a __syncthreads() would drain them).  With the
//    short K of this model (768..3072) a one-deep prefetch is pure load latency.
//  * 128-byte LDS rows, 16-byte chunks XOR-swizzled (chunk ^= (row>>1)&7): every ds_read_b128
//    fragment read is bank-conflict free; the swizzle is applied on the per-lane SOURCE address
//    (the LDS-DMA destination is lane-linear) and on the read.
//  * tile order: bijective XCD remap, then groups of 4 row-tiles x all column tiles, so the 32
//    workgroups that share an XCD's L2 work on a 4 x 8 patch of tiles (A and B panels L2 resident).
//  * the weight tile is the MFMA A operand and the activation tile the B operand, so a lane holds
//    4 consecutive output columns; the epilogue restages the wave's 64x64 f32 block through LDS
//    and finishes in row-chunk form (8 consecutive columns per lane): 16-byte bias / residual
//    loads and 16-byte coalesced stores.
#include "gemm_nt_impl.h"

namespace {

// WM = 4: 8 waves of 64x64 (512 threads).  WM = 2: 4 waves of 128x64 (256 threads): 25 % fewer LDS
// fragment reads per MFMA and half the waves per barrier, 256 registers per wave available.
template <int EPI, int BK, int STAGES, int WM, int WN, int MT_, int SCHED>
__global__ __launch_bounds__(WM * WN * 64, (Cfg<BK, STAGES, WM, WN, MT_>::WG_PER_CU * WM * WN / 4))
void gemm_nt_kernel(GemmNtArgs p) {
  using C = Cfg<BK, STAGES, WM, WN, MT_>;
  constexpr int BM = C::BM, BN = C::BN;
  constexpr int A_BYTES = C::A_BYTES, STAGE_BYTES = C::STAGE_BYTES, ROW_B = C::ROW_B, CHUNKS = C::CHUNKS;
  constexpr int RPI = C::ROWS_PER_INSTR, NA = C::NA, NB = C::NB, KS = C::KS;
  constexpr int NWAVES = C::NWAVES, MT = C::MT, WROWS = C::MT * 16;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // device-side row count (sync-free joint pass): the tile grid is laid over the rows that are really there, exactly as a
  // launch of that size would lay it (valid tiles evenly spread over the XCDs); the surplus workgroups return at once
  const int Mv = rows_clamped(p.M, p.rows_dev);
  const int tiles_m = (p.rows_dev != nullptr) ? (Mv + BM - 1) / BM : p.tiles_m;
  const int nwg = tiles_m * p.tiles_n;
  if ((int)blockIdx.x >= nwg) return;
#ifdef MVPTR_TIMELINE_BUILD
  // diagnostic: wall-clock (100 MHz s_memrealtime) start / loop-end / end of every workgroup
  unsigned long long tl_start, tl_loop, tl_end;
  asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tl_start)::"memory");
#endif
  const int t = xcd_remap(blockIdx.x, nwg);
  // order of the logical tiles (an XCD owns a contiguous run of them): column tiles in chunks of
  // group_n; inside a chunk, groups of group_m row tiles x the chunk's columns, row tile fastest.
  // A chunk narrower than the matrix keeps that part of B in the XCD's L2 while its rows stream by.
  const int chunk_full = tiles_m * p.group_n;
  const int chunk = t / chunk_full;
  const int cn0 = chunk * p.group_n;
  const int cn = min(p.group_n, p.tiles_n - cn0);
  const int tc = t - chunk * chunk_full;
  const int gsz = p.group_m * cn;
  const int grp = tc / gsz;
  const int first_m = grp * p.group_m;
  const int gm = min(p.group_m, tiles_m - first_m);
  const int in_g = tc - grp * gsz;
  const int tm = first_m + in_g % gm;
  const int tn = cn0 + in_g / gm;
  const int m0 = tm * BM, n0 = tn * BN;
  const int rows_a = min(BM, Mv - m0);
  const int rows_b = min(BN, p.N - n0);

  const __amdgpu_buffer_rsrc_t rsA =
      make_rsrc(p.A + (int64_t)m0 * p.lda, (uint32_t)(((int64_t)(rows_a - 1) * p.lda + p.K) * 2));
  const __amdgpu_buffer_rsrc_t rsB =
      make_rsrc(p.B + (int64_t)n0 * p.ldb, (uint32_t)(((int64_t)(rows_b - 1) * p.ldb + p.K) * 2));

  // split-K launches: this workgroup's slice of the reduction index (whole K otherwise)
  const int k_begin = (p.k_split_len > 0) ? (int)blockIdx.y * p.k_split_len : 0;
  const int k_end = (p.k_split_len > 0) ? min(p.K, k_begin + p.k_split_len) : p.K;
  // staging: a wave instruction fills RPI LDS rows (1 KiB, lane-linear); NA per wave for A, NB for B
  uint32_t offA[NA], offB[NB];
  int kcA[NA], kcB[NB];
#pragma unroll
  for (int i = 0; i < NA; ++i) {
    const int row = (i * NWAVES + wave) * RPI + lane / CHUNKS;
    const int c = (lane % CHUNKS) ^ swz_row(row, CHUNKS);
    kcA[i] = c * 8;
    offA[i] = (uint32_t)(row * p.lda * 2 + c * 16);
  }
#pragma unroll
  for (int i = 0; i < NB; ++i) {
    const int row = (i * NWAVES + wave) * RPI + lane / CHUNKS;
    const int c = (lane % CHUNKS) ^ swz_row(row, CHUNKS);
    kcB[i] = c * 8;
    offB[i] = (uint32_t)(row * p.ldb * 2 + c * 16);
  }
  auto stage = [&](int buf, int k0) {
    char* la = lds + buf * STAGE_BYTES;
    char* lb = la + A_BYTES;
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      const uint32_t va = (k0 + kcA[i] < k_end) ? offA[i] + (uint32_t)k0 * 2 : MVPTR_OOB;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, LDS_PTR(la + (i * NWAVES + wave) * 1024), 16, va, 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      const uint32_t vb = (k0 + kcB[i] < k_end) ? offB[i] + (uint32_t)k0 * 2 : MVPTR_OOB;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, LDS_PTR(lb + (i * NWAVES + wave) * 1024), 16, vb, 0, 0, 0);
    }
  };

  const int wm = wave / WN, wn = wave % WN;
  const int c16 = lane & 15, q4 = lane >> 4;
  uint32_t fx[MT][KS], fw[4][KS];
#pragma unroll
  for (int i = 0; i < MT; ++i) {
    const int rx = wm * WROWS + i * 16 + c16;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) fx[i][ks] = rx * ROW_B + (((ks * 4 + q4) ^ swz_row(rx, CHUNKS)) << 4);
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int rw = wn * 64 + i * 16 + c16;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) fw[i][ks] = rw * ROW_B + (((ks * 4 + q4) ^ swz_row(rw, CHUNKS)) << 4);
  }

  f32x4 acc[4][MT];  // [nt][mt]
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < MT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int nk = (k_end - k_begin + BK - 1) / BK;
  constexpr int LPS = NA + NB;  // loads per stage per thread
  // prologue: STAGES-1 stages in flight
#pragma unroll
  for (int s = 0; s < STAGES - 1; ++s)
    if (s < nk) stage(s, k_begin + s * BK);
  int buf = 0;
#ifdef MVPTR_STAMP_BUILD
  unsigned long long t_wait = 0, t_issue = 0, t_lds = 0, t_mfma = 0, ts0, ts1;
#define STAMP(v) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(v)::"memory")
  const unsigned long long t_begin = __builtin_readcyclecounter();
#endif
  for (int kt = 0; kt < nk; ++kt) {
#ifdef MVPTR_STAMP_BUILD
    STAMP(ts0);
#endif
    // stage kt has landed once only the loads of the (up to STAGES-2) younger stages remain
    const int younger = min(STAGES - 2, nk - 1 - kt);
    if (younger >= 3)
      asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(3 * LPS) : "memory");
    else if (younger == 2)
      asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(2 * LPS) : "memory");
    else if (younger == 1)
      asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(LPS) : "memory");
    else
      asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
#ifdef MVPTR_STAMP_BUILD
    STAMP(ts1);
    t_wait += ts1 - ts0;
#endif
    const char* la = lds + buf * STAGE_BYTES;
    const char* lb = la + A_BYTES;
    // fragment reads of the first k-substep are issued before the next stage's address math and
    // LDS-DMA issue, so that work overlaps the LDS read latency instead of preceding it
    bf16x8 xf[MT], wf[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) wf[i] = *reinterpret_cast<const bf16x8*>(lb + fw[i][0]);
#pragma unroll
    for (int i = 0; i < MT; ++i) xf[i] = *reinterpret_cast<const bf16x8*>(la + fx[i][0]);
    // SCHED 1 (diagnostic A/B, MVPTR_GEMM_CFG=f): the fragments of the second k-substep are requested here too, so the LDS
    // streams all 24 reads of the stage while the first 32 MFMAs run, instead of stalling on 12 reads between the two halves
    bf16x8 xf1[SCHED == 1 ? MT : 1], wf1[SCHED == 1 ? 4 : 1];
    if constexpr (SCHED == 1 && KS == 2) {
#pragma unroll
      for (int i = 0; i < 4; ++i) wf1[i] = *reinterpret_cast<const bf16x8*>(lb + fw[i][1]);
#pragma unroll
      for (int i = 0; i < MT; ++i) xf1[i] = *reinterpret_cast<const bf16x8*>(la + fx[i][1]);
    }
    if (kt + STAGES - 1 < nk) {
      int nb = buf + STAGES - 1;
      if (nb >= STAGES) nb -= STAGES;
      stage(nb, k_begin + (kt + STAGES - 1) * BK);
    }
#ifdef MVPTR_STAMP_BUILD
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_memtime %0" : "=s"(ts0)::"memory");
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    t_issue += ts0 - ts1;
    STAMP(ts1);
    t_lds += ts1 - ts0;
    __builtin_amdgcn_sched_barrier(0);
#endif
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      if (ks > 0) {
        if constexpr (SCHED == 1 && KS == 2) {
#pragma unroll
          for (int i = 0; i < 4; ++i) wf[i] = wf1[i];
#pragma unroll
          for (int i = 0; i < MT; ++i) xf[i] = xf1[i];
        } else {
#pragma unroll
          for (int i = 0; i < 4; ++i) wf[i] = *reinterpret_cast<const bf16x8*>(lb + fw[i][ks]);
#pragma unroll
          for (int i = 0; i < MT; ++i) xf[i] = *reinterpret_cast<const bf16x8*>(la + fx[i][ks]);
        }
      }
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int nt = 0; nt < 4; ++nt)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
          acc[nt][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[nt], xf[mt], acc[nt][mt], 0, 0, 0);
      __builtin_amdgcn_s_setprio(0);
    }
#ifdef MVPTR_STAMP_BUILD
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");  // let the last MFMA drain before stamping
    STAMP(ts0);
    t_mfma += ts0 - ts1;
#endif
    buf = (buf + 1 == STAGES) ? 0 : buf + 1;
  }
#ifdef MVPTR_STAMP_BUILD
  if (p.stamps != nullptr && tid == 0) {
    unsigned long long* o = p.stamps + (size_t)blockIdx.x * 8;
    o[0] = t_wait;
    o[1] = t_issue;
    o[2] = t_lds;
    o[3] = t_mfma;
    o[4] = (unsigned long long)nk;
  }
#endif

  // ------------------------------------------------------------------ epilogue
#ifdef MVPTR_TIMELINE_BUILD
  asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tl_loop)::"memory");
#endif
#ifdef MVPTR_DIAG_BUILD
  if (p.no_epi) {   // loop-only timing: keep the accumulators alive, write nothing
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < MT; ++j) asm volatile("" ::"v"(acc[i][j]));
    return;
  }
#endif
  __syncthreads();  // every wave is done with the operand ring
  nt_epilogue<EPI, MT, 32>(p, Mv, acc, reinterpret_cast<float*>(lds) + wave * (32 * ST_LD), m0, n0, wm, wn, lane);
#ifdef MVPTR_TIMELINE_BUILD
  asm volatile("s_waitcnt vmcnt(0)\n\ts_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tl_end)::"memory");
  if (p.stamps != nullptr && tid == 0) {
    unsigned long long* o = p.stamps + (size_t)blockIdx.x * 8;
    o[0] = tl_start;
    o[1] = tl_loop;
    o[2] = tl_end;
    o[3] = __builtin_amdgcn_s_getreg((31 << 11) | 4);    // HW_ID
    o[4] = __builtin_amdgcn_s_getreg((31 << 11) | 20);   // XCC_ID
  }
#endif
}

// ---------------------------------------------------------------------------------------------------------------
// Round 5: "ping-pong" main loop for the 256 x 256 x 64 tile (gemm_nt8_kernel).
//
// gemm_nt_kernel's loop runs both waves of a SIMD through the same sequence at the same time — wait for the stage, 12
// fragment reads, 32 MFMAs, 12 reads, 32 MFMAs — so the matrix pipe idles while the pair reads LDS or waits (measured: 3 300
// cycles per K-step against the 2 048 its MFMAs take).  Here a K-step is four phases of 16 MFMAs per wave (one 64 x 32
// quadrant of the wave's 128 x 64 block over K = 64), every phase = a LOAD segment (fragment reads of the sub-tile the
// phase brings in + the LDS-DMA of ONE 16-KiB half tile) and an MFMA segment, a raw s_barrier after each; the four waves of
// row half 1 run ONE BARRIER BEHIND the four of row half 0 (they share SIMDs pairwise: wave w and w + 4), so on every SIMD
// one wave multiplies while its partner reads and issues.  Operand tiles are staged as half tiles (128 rows x 64 k, whole
// 128-byte lines, XOR-swizzled chunks as before): a half tile's buffer is refilled as soon as its last reader is past it —
// the activation halves of K-step j + 2 go out in phases 3 / 4 of step j (their last read is phase 2: the first 64-row
// sub-tile stays in registers for phase 4), the weight halves of step j + 1 in phases 1 / 2 — so 1.5 K-steps are in flight
// inside the same 128 KiB, and ONE counted vmcnt per K-step (phase 4) certifies the next step; it is read a phase later.
//   phase   quadrant (rows, cols)   fragment reads              LDS-DMA issued
//     1     (m0, n0)                4 weight (n0) + 8 act (m0)  weight half 0 of step j + 1
//     2     (m1, n0)                8 act (m1)                  weight half 1 of step j + 1
//     3     (m1, n1)                4 weight (n1)               activation half 0 of step j + 2
//     4     (m0, n1)                —                           activation half 1 of step j + 2; s_waitcnt vmcnt(4)
// Needs N % 256 == 0 and K % 64 == 0 (rows past M read as zeros through the descriptor); epilogue = nt_epilogue.
// MT = 16-row blocks per wave (8: 256-row tiles; 7 / 6 / 5: 224 / 192 / 160 rows — the launch picks the height that wastes the
// fewest CU rounds); a row half holds MT * 16 of its 128 LDS rows, sub-tile m0 = the first (MT + 1) / 2 blocks, m1 = the rest.
template <int EPI, int MT>
__global__ __launch_bounds__(512, 2) void gemm_nt8_kernel(GemmNtArgs p) {
  constexpr int HALF_B = 128 * 128;     // 128 rows x 64 k bf16
  constexpr int SET_B = 4 * HALF_B;     // activation halves 0 1, weight halves 0 1
  constexpr int HROWS = MT * 16, BM = 2 * HROWS;      // rows of a row half / of the tile
  constexpr int MH0 = (MT + 1) / 2;                  // blocks of sub-tile m0
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int Mv = rows_clamped(p.M, p.rows_dev);
  const int tiles_m = (p.rows_dev != nullptr) ? (Mv + BM - 1) / BM : p.tiles_m;
  const int nwg = tiles_m * p.tiles_n;
  if ((int)blockIdx.x >= nwg) return;
  const int t = xcd_remap(blockIdx.x, nwg);
  const int chunk_full = tiles_m * p.group_n;
  const int chunk = t / chunk_full;
  const int cn0 = chunk * p.group_n;
  const int cn = min(p.group_n, p.tiles_n - cn0);
  const int tc = t - chunk * chunk_full;
  const int gsz = p.group_m * cn;
  const int grp = tc / gsz;
  const int first_m = grp * p.group_m;
  const int gm = min(p.group_m, tiles_m - first_m);
  const int in_g = tc - grp * gsz;
  const int tm = first_m + in_g % gm;
  const int tn = cn0 + in_g / gm;
  const int m0 = tm * BM, n0 = tn * 256;
  const int rows_a = min(BM, Mv - m0);
  const u32x4 rsA = make_rsrc_words(p.A + (int64_t)m0 * p.lda, (uint32_t)(((int64_t)(rows_a - 1) * p.lda + p.K) * 2));
  const u32x4 rsB = make_rsrc_words(p.B + (int64_t)n0 * p.ldb, (uint32_t)(((int64_t)255 * p.ldb + p.K) * 2));
  const uint32_t lds0 = lds_addr(lds);

  const int wm = wave >> 2, wn = wave & 3;      // wm = row half = ping-pong group
  const int c16 = lane & 15, q4 = lane >> 4;
  // staging: instruction i (0, 1) of wave w fills rows i * 64 + w * 8 + lane / 8 of a half tile, 16-byte chunk lane % 8
  // (lane-linear KiB); the swizzle term (row >> 1) & 7 = ((lane >> 4) + 4 w) & 7 does not depend on i or on the half
  uint32_t vbA, vbB;
  {
    const int r = wave * 8 + (lane >> 3);
    const int c = (lane & 7) ^ ((r >> 1) & 7);
    vbA = (uint32_t)(r * p.lda * 2 + c * 16);
    vbB = (uint32_t)(r * p.ldb * 2 + c * 16);
  }
  const uint32_t a64 = (uint32_t)(64 * p.lda * 2), b64 = (uint32_t)(64 * p.ldb * 2);
  const uint32_t ahalf = (uint32_t)(HROWS * p.lda * 2);
  // second staging instruction of an activation half: its rows 64 + 8 w ... exist only below HROWS (wave-uniform); the
  // others are requested out of range (zeros, no fetch) so that every wave issues the same number of LDS-DMA per half
  const uint32_t a_hi = (64 + wave * 8 < HROWS) ? a64 : 0x80000000u;
  // half tile hh (0, 1: activation rows 0-127 / 128-255; 2, 3: weight rows) of K-step kt into buffer set `set`
  auto stage_half = [&](int set, int hh, int kt) {
    const uint32_t la = lds0 + (uint32_t)(set * SET_B + hh * HALF_B + wave * 1024);
    const uint32_t k2 = (uint32_t)kt * 128u;
    if (hh < 2) {
      lds_dma16_add(rsA, vbA, (uint32_t)hh * ahalf + k2, la);
      lds_dma16_add(rsA, vbA, (uint32_t)hh * ahalf + a_hi + k2, la + 8192);
    } else {
      lds_dma16_add(rsB, vbB, (uint32_t)(2 * (hh - 2)) * b64 + k2, la);
      lds_dma16_add(rsB, vbB, (uint32_t)(2 * (hh - 2) + 1) * b64 + k2, la + 8192);
    }
  };
  // fragment reads: row i * 16 + c16 of a half tile, chunk (ks * 4 + q4) ^ ((c16 >> 1) & 7): one per-lane offset per
  // k-substep, blocks are immediates of i * 2048 bytes
  uint32_t fx[2];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) fx[ks] = (uint32_t)(c16 * 128 + (((ks * 4 + q4) ^ ((c16 >> 1) & 7)) << 4));
  const uint32_t offX = (uint32_t)(wm * HALF_B);                                   // this wave's activation half
  const uint32_t offW = (uint32_t)((2 + (wn >> 1)) * HALF_B + (wn & 1) * 8192);    // its 64 weight rows

  f32x4 acc[4][MT];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < MT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  bf16x8 xf[MT][2], wf[4][2];      // [block][k-substep]
  auto read_x = [&](const char* base, int mh) {
#pragma unroll
    for (int i = (mh ? MH0 : 0); i < (mh ? MT : MH0); ++i)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
        xf[i][ks] = *reinterpret_cast<const bf16x8*>(base + offX + i * 2048 + fx[ks]);
  };
  auto read_w = [&](const char* base, int nh) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
        wf[nh * 2 + i][ks] = *reinterpret_cast<const bf16x8*>(base + offW + (nh * 2 + i) * 2048 + fx[ks]);
  };
  auto mma = [&](int mh, int nh) {
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int nt = nh * 2; nt < nh * 2 + 2; ++nt)
#pragma unroll
        for (int mt = (mh ? MH0 : 0); mt < (mh ? MT : MH0); ++mt)
          acc[nt][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[nt][ks], xf[mt][ks], acc[nt][mt], 0, 0, 0);
    __builtin_amdgcn_s_setprio(0);
  };
#define NT8_BARRIER()                         \
  do {                                        \
    __builtin_amdgcn_sched_barrier(0);        \
    asm volatile("s_barrier" ::: "memory");   \
    __builtin_amdgcn_sched_barrier(0);        \
  } while (0)

  const int nk = p.K >> 6;
  // prologue: K-step 0 whole, the activation halves of K-step 1
#pragma unroll
  for (int hh = 0; hh < 4; ++hh) stage_half(0, hh, 0);
  if (nk > 1) {
    stage_half(1, 0, 1);
    stage_half(1, 1, 1);
    asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  } else {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  NT8_BARRIER();
  if (wm == 1) NT8_BARRIER();      // row half 1 runs one barrier behind: its load segments face the other half's MFMA segments
  for (int kt = 0; kt < nk; ++kt) {
    const int set = kt & 1;
    const char* cur = lds + set * SET_B;
    // phase 1
    read_w(cur, 0);
    read_x(cur, 0);
    if (kt + 1 < nk) stage_half(set ^ 1, 2, kt + 1);
    NT8_BARRIER();
    mma(0, 0);
    NT8_BARRIER();
    // phase 2
    read_x(cur, 1);
    if (kt + 1 < nk) stage_half(set ^ 1, 3, kt + 1);
    NT8_BARRIER();
    mma(1, 0);
    NT8_BARRIER();
    // phase 3
    read_w(cur, 1);
    if (kt + 2 < nk) stage_half(set, 0, kt + 2);
    NT8_BARRIER();
    mma(1, 1);
    NT8_BARRIER();
    // phase 4: the next K-step is certified here (everything but the two halves just requested) and read a phase later
    if (kt + 2 < nk) {
      stage_half(set, 1, kt + 2);
      asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    NT8_BARRIER();
    mma(0, 1);
    NT8_BARRIER();
  }
  if (wm == 0) NT8_BARRIER();      // barrier counts of the two halves meet again
#undef NT8_BARRIER
#ifdef MVPTR_DIAG_BUILD
  if (p.no_epi) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < MT; ++j) asm volatile("" ::"v"(acc[i][j]));
    return;
  }
#endif
  __syncthreads();
  nt_epilogue<EPI, MT, 32>(p, Mv, acc, reinterpret_cast<float*>(lds) + wave * (32 * ST_LD), m0, n0, wm, wn, lane);
}

// epilogues the ping-pong kernel is instantiated for: the encoder's (bias / GELU / residual / GELU backward / add) and the bf16-stash
// forms of the GELU pair, so that a gelu_stash = "bf16" A/B run compares stash formats and not kernels
template <int EPI>
constexpr bool kNt8Epi = (EPI <= MVPTR_EPI_ADD) || EPI == MVPTR_EPI_BIAS_GELU_BF16 || EPI == MVPTR_EPI_GELU_BWD_BF16;
template <int EPI>
constexpr bool kGeluFwd = (EPI == MVPTR_EPI_BIAS_GELU || EPI == MVPTR_EPI_BIAS_GELU_BF16);
template <int EPI>
constexpr bool kGeluBwd = (EPI == MVPTR_EPI_GELU_BWD || EPI == MVPTR_EPI_GELU_BWD_BF16);

template <int EPI>
bool nt8_eligible(const GemmNtArgs& a) {
  if ((a.N & 255) || (a.K & 63) || a.K < 128 || a.splits > 1 || a.k_split_len > 0) return false;
  if ((int64_t)256 * a.lda * 2 >= (int64_t)0x7fffffff || (int64_t)256 * a.ldb * 2 >= (int64_t)0x7fffffff) return false;
  return true;
}

template <int EPI, int MT>
int launch_nt8_mt(GemmNtArgs a, hipStream_t s) {
  constexpr int LDS_BYTES = 2 * 4 * 128 * 128;
  constexpr int BM = 2 * MT * 16;
  a.tiles_m = (a.M + BM - 1) / BM;
  a.tiles_n = a.N / 256;
  hipError_t e = hipFuncSetAttribute((const void*)gemm_nt8_kernel<EPI, MT>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
  if (e != hipSuccess) MVPTR_FAIL(MVPTR_HIP_ERROR, "gemm_nt: set LDS size: %s", hipGetErrorString(e));
  const int nwg = a.tiles_m * a.tiles_n;
  constexpr bool kEncoderEpi = (EPI <= MVPTR_EPI_GELU_BWD) || kGeluFwd<EPI> || kGeluBwd<EPI>;
  const bool chunked = kEncoderEpi && a.tiles_n > 4;
  a.group_m = (kGeluFwd<EPI> && chunked) ? 6 : GROUP_M;
  a.group_n = chunked ? (kGeluBwd<EPI> ? 3 : 4) : a.tiles_n;
  if (mvptr_knobs().nt_group[0] > 0) a.group_m = mvptr_knobs().nt_group[0];
  if (mvptr_knobs().nt_group[1] > 0) a.group_n = min(mvptr_knobs().nt_group[1], a.tiles_n);
  if (mvptr_knobs().nt_group[0] > 0 && mvptr_knobs().nt_group[1] <= 0) a.group_n = a.tiles_n;
  hipLaunchKernelGGL((gemm_nt8_kernel<EPI, MT>), dim3(nwg), dim3(512), LDS_BYTES, s, a);
  MVPTR_CHECK_LAUNCH("gemm_nt");
  return MVPTR_OK;
}

// Tile height: 256 CUs run the tiles in rounds, and a launch pays whole rounds — 444 tiles of 256 rows (M = 37 748, N = 768)
// are 1.73 rounds = 2, the same rows as 507 tiles of 224 rows are 1.98 rounds = 2 rounds of tiles that take 7 / 8 of the time.
// The height with the smallest rounds x rows x (a small per-row penalty for the shorter tiles: more weight bytes per
// FLOP) is taken; `mt_force` (diagnostic build, MVPTR_NT_EXP bits 22-25) pins it for A/B runs.
template <int EPI>
int launch_nt8(const GemmNtArgs& a, hipStream_t s) {
  const int Mp = (a.m_plan > 0 && a.m_plan < a.M) ? a.m_plan : a.M;
  const int ncu = nt_num_cus();
  const int64_t tn = a.N / 256;
  static const double kPenalty[9] = {0, 0, 0, 0, 0, 1.10, 1.06, 1.03, 1.0};
  int best = 8;
  double best_cost = 1e30, cost8 = 0.0;
  for (int mt = 8; mt >= (a.full_height ? 8 : 5); --mt) {
    const int64_t tiles = (int64_t)((Mp + 32 * mt - 1) / (32 * mt)) * tn;
    const double rounds = (double)((tiles + ncu - 1) / ncu);
    const double cost = rounds * mt * kPenalty[mt];
    if (mt == 8) cost8 = cost;
    if (cost < best_cost - 1e-9) {
      best_cost = cost;
      best = mt;
    }
  }
  // a partly filled last round runs faster per tile than a full one (fewer CUs share HBM, L2 and the clock), so whole rounds
  // overstate what a shorter tile saves: it is only taken for a predicted saving of 15 % or more (in-step A/B, gpurun_out/r05g:
  // 224-row tiles on the joint stack's N = 768 / 2304 GEMMs — predicted -10 % — cost +0.15 ms per step; the 192- / 160-row
  // tiles of a lone few-row launch — predicted -20 ... -31 % — win 0.4 - 0.5 ms on one-stream steps)
  if (best != 8 && best_cost > 0.85 * cost8) best = 8;
#ifdef MVPTR_DIAG_BUILD
  const int force = (mvptr_knobs().nt_exp >> 22) & 15;
  if (force >= 5 && force <= 8) best = force;
#endif
  switch (best) {
    case 5: return launch_nt8_mt<EPI, 5>(a, s);
    case 6: return launch_nt8_mt<EPI, 6>(a, s);
    case 7: return launch_nt8_mt<EPI, 7>(a, s);
    default: return launch_nt8_mt<EPI, 8>(a, s);
  }
}


template <int EPI, int BK, int STAGES, int WM, int WN, int MT_, int SCHED>
int launch_bk(GemmNtArgs a, hipStream_t s) {
  using C = Cfg<BK, STAGES, WM, WN, MT_>;
  constexpr int LDS_BYTES = C::LDS_BYTES;
  a.tiles_m = (a.M + C::BM - 1) / C::BM;
  a.tiles_n = (a.N + C::BN - 1) / C::BN;
  if ((int64_t)C::BM * a.lda * 2 >= (int64_t)0x7fffffff || (int64_t)C::BN * a.ldb * 2 >= (int64_t)0x7fffffff)
    MVPTR_FAIL(MVPTR_BAD_SHAPE, "gemm_nt: leading dimension too large");
  hipError_t e = hipFuncSetAttribute((const void*)gemm_nt_kernel<EPI, BK, STAGES, WM, WN, MT_, SCHED>,
                                     hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
  if (e != hipSuccess) MVPTR_FAIL(MVPTR_HIP_ERROR, "gemm_nt: set LDS size: %s", hipGetErrorString(e));
  const int nwg = a.tiles_m * a.tiles_n;
  // Tile order (tools/sweep_nt_group.py, profiles/r02_experiments.txt): wide outputs are swept in
  // chunks of 4 column tiles (3 for the GELU-backward epilogue, which also reads an M x N operand),
  // groups of 4 row tiles (6 for the two-output GELU epilogue) inside a chunk.  Against whole-width
  // groups: FFN1 forward 283 -> 228 us at M = 37 748 (-19 %), -6...-12 % at the other row counts,
  // Q/K/V projection -13 % at M = 64 000, the rest within 2 %.  MVPTR_NT_GROUP=gm,gn overrides.
  constexpr bool kEncoderEpi = (EPI <= MVPTR_EPI_GELU_BWD);     // bias / gelu / residual / gelu-backward: the measured shapes
  const bool chunked = kEncoderEpi && C::BN == 256 && a.tiles_n > 4;
  a.group_m = (EPI == MVPTR_EPI_BIAS_GELU && chunked) ? 6 : GROUP_M;
  a.group_n = chunked ? ((EPI == MVPTR_EPI_GELU_BWD) ? 3 : 4) : a.tiles_n;
  if (mvptr_knobs().nt_group[0] > 0) a.group_m = mvptr_knobs().nt_group[0];
  if (mvptr_knobs().nt_group[1] > 0) a.group_n = min(mvptr_knobs().nt_group[1], a.tiles_n);
  if (mvptr_knobs().nt_group[0] > 0 && mvptr_knobs().nt_group[1] <= 0) a.group_n = a.tiles_n;   // "gm" or "gm,0": whole width
  hipLaunchKernelGGL((gemm_nt_kernel<EPI, BK, STAGES, WM, WN, MT_, SCHED>), dim3(nwg, a.splits > 1 ? a.splits : 1), dim3(WM * WN * 64),
                     LDS_BYTES, s, a);
  MVPTR_CHECK_LAUNCH("gemm_nt");
  return MVPTR_OK;
}

template <int EPI>
int launch(const GemmNtArgs& a, hipStream_t s) {
  // Tile configurations (measured on MI355X, profiles/r01_gemm_configs.txt, profiles/r02_experiments.txt,
  // profiles/r03_experiments.txt):
  //   256x256, BK 64 (whole 128-B lines per row), double buffer, 8 waves of 128x64, one workgroup per CU: fewest
  //           L2->LDS bytes per FLOP, loop rate 1.2-1.4 PF/s — the default;
  //   256x128, BK 32, 3-stage ring, 8 waves of 64x64, two workgroups per CU: the second workgroup's loop runs
  //           beside the first one's epilogue, lower loop rate (1.0-1.15 PF/s) — short-K narrow GEMMs whose 256x256
  //           tiles would need more than one round of the 256 CUs (attention-output projection on the big batch);
  //   128x128, BK 32, 4 waves, up to three workgroups per CU — few-row GEMMs (head transforms on the masked rows).
  // Measured and dropped (kernels removed, numbers in the experiment logs): BK 32 rings of 3 / 4 / 5 stages for the
  // 256x256 tile, staggered wave halves, persistent workgroups with next-tile prefetch or deferred epilogues,
  // one-wave-per-SIMD 128x128 wave tiles with a register epilogue.
#ifdef MVPTR_DIAG_BUILD
  const char* env = mvptr_knobs().gemm_cfg;
  if (env[0] != 0) {
    if (env[0] == '8') {                                                  // "8": ping-pong loop (gemm_nt8_kernel) where eligible
      if constexpr (kNt8Epi<EPI>) {
        if (nt8_eligible<EPI>(a)) return launch_nt8<EPI>(a, s);
      }
      return launch_bk<EPI, 64, 2, 2, 4, 8, 0>(a, s);
    }
    if (env[0] == 'n' || env[0] == 'p') {                                  // "n768", "p", "pd": the rejected experiments (diag_gemm.hip)
      const int rc = mvptr_diag_gemm_nt(EPI, a, env, s);
      if (rc != MVPTR_DIAG_NOT_HANDLED) return rc;
      return launch_bk<EPI, 64, 2, 2, 4, 8, 0>(a, s);
    }
    if (env[0] == 'v') return launch_bk<EPI, 32, 3, 2, 2, 8, 0>(a, s);  // "v4": 256x128, FOUR waves of 128x64, two workgroups per CU
    if (env[0] == 's') return launch_bk<EPI, 32, 3, 2, 2, 4, 0>(a, s);  // "s128"
    if (env[0] == 'w') return launch_bk<EPI, 32, 3, 4, 2, 4, 0>(a, s);  // "w4"
    if (env[0] == 'm' && env[1] == '6') return launch_bk<EPI, 64, 2, 2, 4, 6, 0>(a, s);  // "m6": 192 x 256 tiles (tail split A/B)
    if (env[0] == 'm' && env[1] == '4') return launch_bk<EPI, 64, 2, 2, 4, 4, 0>(a, s);  // "m4": 128 x 256
    if (env[0] == 'm' && env[1] == '2') return launch_bk<EPI, 64, 2, 2, 4, 2, 0>(a, s);  // "m2": 64 x 256
    if (env[0] == 'f') return launch_bk<EPI, 64, 2, 2, 4, 8, 1>(a, s);  // "f": all fragment reads of a stage up front (SCHED 1)
    if (env[0] == 't') return launch_bk<EPI, 64, 2, 2, 4, 8, 0>(a, s);  // "t256k"
    MVPTR_FAIL(MVPTR_BAD_ARG, "gemm_nt: unknown MVPTR_GEMM_CFG '%s'", env);
  }
#endif
  const int Mp = (a.m_plan > 0 && a.m_plan < a.M) ? a.m_plan : a.M;     // rows the configuration is chosen for
  const int64_t tiles256 = (int64_t)((Mp + 255) / 256) * ((a.N + 255) / 256);
  // Tile height (round 4).  256x256 tiles over the 256 CUs run in rounds: 129 tiles (M = 10 917, N = 768) use half the chip
  // for one round, 516 (N = 3072) pay a third round for 4 tiles.  192-row tiles (MT_ 6 of the same kernel; ~10 % slower per
  // FLOP: more operand bytes per output) are taken where they save most of a round: measured cold at M = 10 917
  // (profiles/r04_experiments.txt) attention output 45.9 -> 37.0 us, FFN1 100.4 -> 86.1, FFN2 108.1 -> 100.8, Q/K/V dgrad
  // 80.4 -> 75.4; one-stream step 29.4 -> 28.6 ms.  Only for a single under-filled round or a last round that is nearly
  // empty (<= 15 % of the CUs): a launch of 1.4 rounds (M = 30 720, N = 768) looks like a win by the same arithmetic and in
  // a cold replay (211.8 -> 196.0 us) but LOSES inside the step (single-stream model 16.75 -> 17.5 ms): its second round
  // already runs faster per tile.  Splitting the last round off into a second launch of shorter tiles was measured too and
  // is not built (no gain on the joint stack: 444 tiles, 214.6 -> 221.6 us).
  bool nt8_off = false;
#ifdef MVPTR_DIAG_BUILD
  nt8_off = (mvptr_knobs().nt_exp & (1 << 17)) != 0;                      // MVPTR_NT_EXP bit 17: the round-4 kernels (A/B)
#endif
  // Round 5: the ping-pong loop (gemm_nt8_kernel) wherever the shape allows (N % 256 == 0, K % 64 == 0: every encoder GEMM of
  // more than 64 tiles), with its own tile-height rule.  Cold table, M = 37 748 / 64 000 (profiles/r05_experiments.txt):
  // loop-only 141 -> 125 us (Q/K/V), 188 -> 155 (K = 3072), whole kernels -4 ... -17 %; it also replaces the 256 x 128
  // two-workgroup configuration of the narrow short-K GEMMs (attention output on the joint stack: 83.9 -> 73.7 us).
  if constexpr (kNt8Epi<EPI>) {      // the encoder epilogues (the others never meet the shape rule on this model's paths)
    if (!nt8_off && tiles256 > 64 && nt8_eligible<EPI>(a)) return launch_nt8<EPI>(a, s);
  }
  if constexpr (EPI <= MVPTR_EPI_ADD) {
    bool off = false;
#ifdef MVPTR_DIAG_BUILD
    off = (mvptr_knobs().nt_exp & 65536) != 0;                           // MVPTR_NT_EXP bit 16: 256-row tiles only (A/B)
#endif
    const int ncu = nt_num_cus();
    const int last = (int)(tiles256 % ncu);                               // tiles of the last round
    const bool wasteful = tiles256 <= ncu || (last > 0 && last * 100 <= ncu * 15);
    if (!off && tiles256 > 64 && wasteful && !(a.N <= 768 && a.K <= 768 && tiles256 > 256)) {
      const int64_t tn = (a.N + 255) / 256;
      const double r256 = (double)((tiles256 + ncu - 1) / ncu);
      const double r192 = (double)(((int64_t)((Mp + 191) / 192) * tn + ncu - 1) / ncu) * 0.75 * kShortTilePenalty;
      if (r192 < 0.92 * r256) return launch_bk<EPI, 64, 2, 2, 4, 6, 0>(a, s);
    }
  }
  // few-row GEMMs (head transforms on the masked rows: M ~ 3 k, N = 768) would give a 256x256 tile to
  // a quarter of the CUs or fewer: 128x128 tiles, 4 waves, up to three workgroups per CU
  // (35 vs 74 us at M = 3000, N = 768, K = 3072; at M = 11 k the big tile still wins, 74 vs 87 us)
  if (tiles256 <= 64 && Mp > 128) return launch_bk<EPI, 32, 3, 2, 2, 4, 0>(a, s);
  if (a.N <= 768 && a.K <= 768 && tiles256 > 256) return launch_bk<EPI, 32, 3, 4, 2, 4, 0>(a, s);
  return launch_bk<EPI, 64, 2, 2, 4, 8, 0>(a, s);
}

}  // namespace

namespace {
__global__ __launch_bounds__(256) void ce_finalize_kernel(const float* part, int part_ld, const float* lab_logit,
                                                          const int64_t* labels, float* loss_row, float* lse_row, int M, int V) {
  // one wave per row: combine the per-strip (max, sum) partials
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= M) return;
  float mx = -1e30f, sm = 0.f;
  for (int i = lane; i < part_ld; i += 64) {
    const float m2 = part[((int64_t)row * part_ld + i) * 2], s2 = part[((int64_t)row * part_ld + i) * 2 + 1];
    const float mm = fmaxf(mx, m2);
    sm = sm * __expf(mx - mm) + s2 * __expf(m2 - mm);
    mx = mm;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float m2 = __shfl_xor(mx, o), s2 = __shfl_xor(sm, o);
    const float mm = fmaxf(mx, m2);
    sm = sm * __expf(mx - mm) + s2 * __expf(m2 - mm);
    mx = mm;
  }
  if (lane == 0) {
    const float lse = mx + logf(sm);
    lse_row[row] = lse;
    const int64_t lab = labels[row];
    loss_row[row] = (lab >= 0 && lab < V) ? (lse - lab_logit[row]) : 0.f;
  }
}

int decoder_args(GemmNtArgs& a, const void* h, int64_t ldh, const void* W, int64_t ldw, const float* bias,
                 const int64_t* labels, int M, int V, int K, const char* who) {
  if (M <= 0 || V <= 0 || K <= 0) MVPTR_FAIL(MVPTR_BAD_SHAPE, "%s: M, V, K must be > 0", who);
  if ((K & 7) || (ldh & 7) || (ldw & 7) || ldh < K || ldw < K) MVPTR_FAIL(MVPTR_BAD_ALIGN, "%s: K, ldh, ldw must be multiples of 8 and >= K", who);
  if (!h || !W || !labels) MVPTR_FAIL(MVPTR_BAD_ARG, "%s: NULL argument", who);
  if (((uintptr_t)h & 15) || ((uintptr_t)W & 15)) MVPTR_FAIL(MVPTR_BAD_ALIGN, "%s: h and W must be 16-byte aligned", who);
  memset(&a, 0, sizeof(a));
  a.A = (const __bf16*)h;
  a.B = (const __bf16*)W;
  a.lda = ldh;
  a.ldb = ldw;
  a.M = M;
  a.N = V;
  a.K = K;
  a.bias = bias;
  a.drop = make_dropdev(nullptr);
  a.labels = labels;
  a.vec_bias_ok = (bias && (((uintptr_t)bias & 15) == 0)) ? 1 : 0;
  return MVPTR_OK;
}
}  // namespace

// Vocabulary decoder + CrossEntropyLoss without the [M, V] f32 logits in HBM
// (modeling_bert.py:513-516 + the loss of modeling_vlbert.py:1112-1125,1245-1249).
// Forward: logits tile by tile in the GEMM epilogue, reduced to per-row (max, sum exp) partials per
// 64-column strip plus the logit at the label; a small kernel folds the partials into lse / loss.
extern "C" int mvptr_decoder_ce_fwd(const void* h, int64_t ldh, const void* W, int64_t ldw, const float* bias,
                                    const int64_t* labels, int M, int V, int K, float* part, float* lab_logit,
                                    float* loss_row, float* lse_row, void* stream) {
  GemmNtArgs a;
  const int rc = decoder_args(a, h, ldh, W, ldw, bias, labels, M, V, K, "decoder_ce_fwd");
  if (rc != MVPTR_OK) return rc;
  if (!part || !lab_logit || !loss_row || !lse_row) MVPTR_FAIL(MVPTR_BAD_ARG, "decoder_ce_fwd: NULL output");
  a.part = part;
  a.lab_logit = lab_logit;
  a.part_ld = (V + 63) / 64;
  a.out0 = part;  // not written by this epilogue
  a.ldc = 8;
  const int rc2 = launch<EPI_CE_PART>(a, (hipStream_t)stream);
  if (rc2 != MVPTR_OK) return rc2;
  hipLaunchKernelGGL(ce_finalize_kernel, dim3((M + 3) / 4), dim3(256), 0, (hipStream_t)stream, part, a.part_ld,
                     lab_logit, labels, loss_row, lse_row, M, V);
  MVPTR_CHECK_LAUNCH("decoder_ce_fwd");
  return MVPTR_OK;
}

// Backward: the logits are recomputed by the same GEMM and leave its epilogue as
// d = (softmax - onehot) * scale in bf16 [M, ld_d] (columns V..Vpad-1 zero), the operand of the
// data- and weight-gradient GEMMs.
extern "C" int mvptr_decoder_ce_bwd(const void* h, int64_t ldh, const void* W, int64_t ldw, const float* bias,
                                    const int64_t* labels, const float* lse_row, const float* scale, int M, int V,
                                    int K, void* dlogits, int64_t ld_d, int Vpad, void* stream) {
  GemmNtArgs a;
  const int rc = decoder_args(a, h, ldh, W, ldw, bias, labels, M, V, K, "decoder_ce_bwd");
  if (rc != MVPTR_OK) return rc;
  if (!lse_row || !scale || !dlogits || Vpad < V || ld_d < Vpad) MVPTR_FAIL(MVPTR_BAD_ARG, "decoder_ce_bwd: bad argument");
  a.lse = lse_row;
  a.scale = scale;
  a.out0 = dlogits;
  a.ldc = ld_d;
  a.n_store = Vpad;
  a.vec_out_ok = ((ld_d % 8 == 0) && (((uintptr_t)dlogits & 15) == 0)) ? 1 : 0;
  return launch<EPI_CE_BWD>(a, (hipStream_t)stream);
}

// LayerNorm folded into the neighbouring GEMMs (inference path; gemm_nt_impl.h EPI_FOLD_* / EPI_RESID_LN).  256 x 256 tiles of
// gemm_nt8_kernel only: N % 256 == 0, K % 64 == 0, K >= 128 (every encoder GEMM of a hidden size that is a multiple of 256).
extern "C" int mvptr_gemm_nt_ln(const void* A, int64_t lda, const void* B, int64_t ldb, int M, int N, int K, int mode, const float* bias,
                                const void* aux, int64_t ld_aux, const float* stats, const float* colsum, const float* gamma,
                                const float* beta, void* out, int64_t ldc, float* row_partials, void* stream) {
  if (M <= 0 || N <= 0 || K <= 0) MVPTR_FAIL(MVPTR_BAD_SHAPE, "gemm_nt_ln: M, N, K must be > 0");
  if ((K & 7) || (lda & 7) || (ldb & 7) || lda < K || ldb < K) MVPTR_FAIL(MVPTR_BAD_ALIGN, "gemm_nt_ln: K, lda, ldb must be multiples of 8 and >= K");
  if (!A || !B || !out || !bias || ((uintptr_t)A & 15) || ((uintptr_t)B & 15) || ((uintptr_t)out & 15) || (ldc & 7) || ldc < N)
    MVPTR_FAIL(MVPTR_BAD_ALIGN, "gemm_nt_ln: NULL / unaligned pointer or ldc");
  if (mode < 0 || mode > 2) MVPTR_FAIL(MVPTR_BAD_ARG, "gemm_nt_ln: mode must be MVPTR_LN_FOLD_BIAS, _FOLD_GELU or _RESID_STATS");
  if (mode != MVPTR_LN_RESID_STATS && (!stats || !colsum)) MVPTR_FAIL(MVPTR_BAD_ARG, "gemm_nt_ln: the folded modes need stats and colsum");
  if (mode == MVPTR_LN_RESID_STATS) {
    if (!aux || !row_partials || ld_aux < N || (ld_aux & 7) || ((uintptr_t)aux & 15)) MVPTR_FAIL(MVPTR_BAD_ARG, "gemm_nt_ln: residual rows / partials missing or unaligned");
    if (stats && (!gamma || !beta)) MVPTR_FAIL(MVPTR_BAD_ARG, "gemm_nt_ln: a pre-LayerNorm residual needs gamma and beta");
  }
  GemmNtArgs a;
  memset(&a, 0, sizeof(a));
  a.A = (const __bf16*)A;
  a.B = (const __bf16*)B;
  a.lda = lda;
  a.ldb = ldb;
  a.M = M;
  a.N = N;
  a.K = K;
  a.bias = bias;
  a.aux = (const __bf16*)aux;
  a.ld_aux = ld_aux;
  a.out0 = out;
  a.ldc = ldc;
  a.drop = make_dropdev(nullptr);
  a.vec_out_ok = 1;
  a.vec_aux_ok = 1;
  a.vec_bias_ok = (((uintptr_t)bias & 15) == 0) ? 1 : 0;
  a.splits = 1;
  a.ln_stats = stats;
  a.ln_c = colsum;
  a.ln_gamma = gamma;
  a.ln_beta = beta;
  a.part = row_partials;
  a.part_ld = N / 64;
  if (!nt8_eligible<MVPTR_EPI_BIAS>(a)) MVPTR_FAIL(MVPTR_BAD_SHAPE, "gemm_nt_ln: needs N %% 256 == 0, K %% 64 == 0, K >= 128 (N = %d, K = %d)", N, K);
  switch (mode) {
    case MVPTR_LN_FOLD_BIAS: return launch_nt8_mt<EPI_FOLD_BIAS, 8>(a, (hipStream_t)stream);
    case MVPTR_LN_FOLD_GELU: return launch_nt8_mt<EPI_FOLD_GELU, 8>(a, (hipStream_t)stream);
    default: return launch_nt8_mt<EPI_RESID_LN, 8>(a, (hipStream_t)stream);
  }
}

extern "C" int mvptr_gemm_nt(const void* A, int64_t lda, const void* B, int64_t ldb, int M, int N,
                             int K, int epilogue, const float* bias, const void* aux,
                             int64_t ld_aux, void* out0, void* out1, int64_t ldc, float* vec_out,
                             const mvptr_dropout* drop, void* stream) {
  return mvptr_gemm_nt_rows(A, lda, B, ldb, M, N, K, epilogue, bias, aux, ld_aux, out0, out1, ldc, vec_out, drop, nullptr, 0, stream);
}

int mvptr_gemm_nt_rows(const void* A, int64_t lda, const void* B, int64_t ldb, int M, int N, int K, int epilogue, const float* bias,
                       const void* aux, int64_t ld_aux, void* out0, void* out1, int64_t ldc, float* vec_out, const mvptr_dropout* drop,
                       const int* rows_dev, int M_plan, void* stream, int full_height) {
  if (M <= 0 || N <= 0 || K <= 0) MVPTR_FAIL(MVPTR_BAD_SHAPE, "gemm_nt: M,N,K must be > 0");
  if ((K & 7) || (lda & 7) || (ldb & 7))
    MVPTR_FAIL(MVPTR_BAD_ALIGN, "gemm_nt: K, lda, ldb must be multiples of 8 (K=%d lda=%ld ldb=%ld)",
               K, (long)lda, (long)ldb);
  if (((uintptr_t)A & 15) || ((uintptr_t)B & 15))
    MVPTR_FAIL(MVPTR_BAD_ALIGN, "gemm_nt: A and B must be 16-byte aligned");
  if (lda < K || ldb < K) MVPTR_FAIL(MVPTR_BAD_SHAPE, "gemm_nt: lda/ldb smaller than K");
  if (out0 == nullptr) MVPTR_FAIL(MVPTR_BAD_ARG, "gemm_nt: out0 is NULL");
  GemmNtArgs a;
  memset(&a, 0, sizeof(a));
  a.A = (const __bf16*)A;
  a.B = (const __bf16*)B;
  a.lda = lda;
  a.ldb = ldb;
  a.M = M;
  a.N = N;
  a.K = K;
  a.rows_dev = rows_dev;
  a.m_plan = M_plan;
  a.full_height = full_height;
  a.bias = bias;
  a.aux = (const __bf16*)aux;
  a.ld_aux = ld_aux;
  a.out0 = out0;
  a.out1 = out1;
  a.ldc = ldc;
  a.vec_out = vec_out;
  a.drop = make_dropdev(drop);
  a.stamps = nullptr;
  a.splits = 1;
  a.k_split_len = 0;
  a.slab_stride = 0;
  a.labels = nullptr;
  a.lse = a.scale = nullptr;
  a.part = a.lab_logit = nullptr;
  a.part_ld = a.n_store = 0;
  a.ln_stats = a.ln_c = a.ln_gamma = a.ln_beta = nullptr;
  const MvptrKnobs& kn = mvptr_knobs();
  a.stash_temporal = (kn.nt_exp & 512) ? 1 : 0;
  if ((kn.nt_exp >> 19) & 7) a.stash_temporal = 1 + ((kn.nt_exp >> 19) & 7);      // bits 19-21: 1 = sc1, 2 = sc0 sc1, 3 = nt (buffer store)
  a.no_epi = (kn.nt_exp & 1024) ? 1 : 0;
  a.epi_ablate = (kn.nt_exp >> 26) & 31;
  a.store_mode = (kn.nt_exp >> 13) & 7;
#if defined(MVPTR_STAMP_BUILD) || defined(MVPTR_TIMELINE_BUILD)
  a.stamps = (unsigned long long*)kn.stamps;
#endif
  a.tiles_m = a.tiles_n = 0;  // set per tile configuration in launch_bk
  const int esz = (epilogue == MVPTR_EPI_F32) ? 4 : 2;
  bool vo = (ldc % 8 == 0) && (((uintptr_t)out0 & 15) == 0);
  if (out1) vo = vo && (((uintptr_t)out1 & 15) == 0);
  (void)esz;
  a.vec_out_ok = vo ? 1 : 0;
  a.vec_aux_ok = (aux && (ld_aux % 8 == 0) && (((uintptr_t)aux & 15) == 0)) ? 1 : 0;
  a.vec_bias_ok = (bias && (((uintptr_t)bias & 15) == 0)) ? 1 : 0;
  hipStream_t s = (hipStream_t)stream;
  switch (epilogue) {
    case MVPTR_EPI_BIAS:
      return launch<MVPTR_EPI_BIAS>(a, s);
    case MVPTR_EPI_BIAS_GELU:
      if (!out1) MVPTR_FAIL(MVPTR_BAD_ARG, "gemm_nt: EPI_BIAS_GELU needs out1");
      return launch<MVPTR_EPI_BIAS_GELU>(a, s);
    case MVPTR_EPI_BIAS_RESID:
      if (!aux) MVPTR_FAIL(MVPTR_BAD_ARG, "gemm_nt: EPI_BIAS_RESID needs aux");
      return launch<MVPTR_EPI_BIAS_RESID>(a, s);
    case MVPTR_EPI_GELU_BWD:
      if (!aux) MVPTR_FAIL(MVPTR_BAD_ARG, "gemm_nt: EPI_GELU_BWD needs aux");
      return launch<MVPTR_EPI_GELU_BWD>(a, s);
    case MVPTR_EPI_ADD:
      return launch<MVPTR_EPI_ADD>(a, s);
    case MVPTR_EPI_F32:
      return launch<MVPTR_EPI_F32>(a, s);
    case MVPTR_EPI_BIAS_TANH:
      return launch<MVPTR_EPI_BIAS_TANH>(a, s);
    case MVPTR_EPI_BIAS_GELU_BF16:
      if (!out1) MVPTR_FAIL(MVPTR_BAD_ARG, "gemm_nt: EPI_BIAS_GELU_BF16 needs out1");
      return launch<MVPTR_EPI_BIAS_GELU_BF16>(a, s);
    case MVPTR_EPI_GELU_BWD_BF16:
      if (!aux) MVPTR_FAIL(MVPTR_BAD_ARG, "gemm_nt: EPI_GELU_BWD_BF16 needs aux");
      return launch<MVPTR_EPI_GELU_BWD_BF16>(a, s);
    default:
      MVPTR_FAIL(MVPTR_BAD_ARG, "gemm_nt: unknown epilogue %d", epilogue);
  }
}


// Split-K form for few-row, long-K products (the data gradient of the vocabulary decoder: dh[M, 768] = dlogits[M, 30528] W,
// M = the ~3 k scored rows of a step): 138 tiles of 128 x 128 cannot fill 256 CUs and each would loop over 954 K-steps.
// The reduction index is cut into `splits` slices; workgroup (tile, z) multiplies slice z and writes its f32 partial tile
// into slab z (plain stores, no atomics: the caller adds the slabs in order, deterministically).
extern "C" int mvptr_gemm_nt_splitk(const void* A, int64_t lda, const void* B, int64_t ldb, int M, int N, int K, int splits,
                                    float* slabs, int64_t ldc, void* stream) {
  if (M <= 0 || N <= 0 || K <= 0 || splits < 1 || splits > 64) MVPTR_FAIL(MVPTR_BAD_SHAPE, "gemm_nt_splitk: M,N,K > 0, 1 <= splits <= 64");
  if ((K & 7) || (lda & 7) || (ldb & 7) || lda < K || ldb < K) MVPTR_FAIL(MVPTR_BAD_ALIGN, "gemm_nt_splitk: K, lda, ldb must be multiples of 8 and >= K");
  if (!A || !B || !slabs || ((uintptr_t)A & 15) || ((uintptr_t)B & 15)) MVPTR_FAIL(MVPTR_BAD_ALIGN, "gemm_nt_splitk: NULL or unaligned pointer");
  if (ldc < N) MVPTR_FAIL(MVPTR_BAD_SHAPE, "gemm_nt_splitk: ldc < N");
  GemmNtArgs a;
  memset(&a, 0, sizeof(a));
  a.A = (const __bf16*)A;
  a.B = (const __bf16*)B;
  a.lda = lda;
  a.ldb = ldb;
  a.M = M;
  a.N = N;
  a.K = K;
  a.out0 = slabs;
  a.ldc = ldc;
  a.drop = make_dropdev(nullptr);
  a.vec_out_ok = ((ldc % 4 == 0) && (((uintptr_t)slabs & 15) == 0) && ((((int64_t)M * ldc) & 3) == 0)) ? 1 : 0;
  a.splits = splits;
  a.k_split_len = ((K + splits - 1) / splits + 63) / 64 * 64;      // whole K-steps of every tile configuration
  a.slab_stride = (int64_t)M * ldc;
  return launch<MVPTR_EPI_F32>(a, (hipStream_t)stream);
}
