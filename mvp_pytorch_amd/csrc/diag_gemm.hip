// diag_gemm.hip — forward / data-gradient GEMM experiments that were MEASURED AND NOT ADOPTED (rounds 3-4), kept out of the
// product's hottest file.  Compiled into libmvptr_hip_diag.so only (`make diag`, -DMVPTR_DIAG_BUILD); the
// product build of this file is empty.  Reached through MVPTR_GEMM_CFG = n768 | p | pd (gemm_nt.hip launch()).
#include "gemm_nt_impl.h"
#ifdef MVPTR_DIAG_BUILD

namespace {
// EXPERIMENT (diagnostic build only, MVPTR_GEMM_CFG=n768): the row-owning tile a LayerNorm-in-the-epilogue GEMM would
// need (north_star "fused LayerNorm"): 128 rows x ALL 768 output columns per workgroup, eight waves of
// 128 x 96 (8 x 6 blocks of v_mfma_f32_16x16x32_bf16 = 192 accumulator registers), BK 32, double-buffered
// (128 + 768) x 64 B = 56 KiB stages (a BK 64 stage would be 112 KiB: the second buffer does not fit).  Epilogue: bias +
// residual straight from the accumulators (8-byte stores; no LayerNorm): the point is the MAIN LOOP of this tile
// shape against the default 256 x 256 tile, measured by tools/exp_rowtile.py with and without epilogues.
__global__ __launch_bounds__(512, 2) void gemm_nt_rowtile_kernel(GemmNtArgs p) {
  constexpr int BM = 128, BN = 768, BK = 32, ROW_B = 64, CHUNKS = 4, RPI = 16, NWAVES = 8, MT = 8, NT = 6;
  constexpr int A_BYTES = BM * BK * 2, STAGE_BYTES = (BM + BN) * BK * 2, NB = BN / RPI / NWAVES;   // 6 B pieces + 1 A piece per wave
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int m0 = blockIdx.x * BM;
  const int rows_a = min(BM, p.M - m0), rows_b = min(BN, p.N);
  const __amdgpu_buffer_rsrc_t rsA = make_rsrc(p.A + (int64_t)m0 * p.lda, (uint32_t)(((int64_t)(rows_a - 1) * p.lda + p.K) * 2));
  const __amdgpu_buffer_rsrc_t rsB = make_rsrc(p.B, (uint32_t)(((int64_t)(rows_b - 1) * p.ldb + p.K) * 2));
  uint32_t offA, offB[NB];
  int kcA, kcB[NB];
  {
    const int row = wave * RPI + lane / CHUNKS;
    const int c = (lane % CHUNKS) ^ swz_row(row, CHUNKS);
    kcA = c * 8;
    offA = (uint32_t)(row * p.lda * 2 + c * 16);
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
    const uint32_t va = (k0 + kcA < p.K) ? offA + (uint32_t)k0 * 2 : MVPTR_OOB;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, LDS_PTR(la + wave * 1024), 16, va, 0, 0, 0);
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      const uint32_t vb = (k0 + kcB[i] < p.K) ? offB[i] + (uint32_t)k0 * 2 : MVPTR_OOB;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, LDS_PTR(lb + (i * NWAVES + wave) * 1024), 16, vb, 0, 0, 0);
    }
  };
  const int c16 = lane & 15, q4 = lane >> 4;
  uint32_t fx[MT], fw[NT];
#pragma unroll
  for (int i = 0; i < MT; ++i) {
    const int rx = i * 16 + c16;
    fx[i] = rx * ROW_B + ((q4 ^ swz_row(rx, CHUNKS)) << 4);
  }
#pragma unroll
  for (int i = 0; i < NT; ++i) {
    const int rw = wave * 96 + i * 16 + c16;
    fw[i] = rw * ROW_B + ((q4 ^ swz_row(rw, CHUNKS)) << 4);
  }
  f32x4 acc[NT][MT];
#pragma unroll
  for (int i = 0; i < NT; ++i)
#pragma unroll
    for (int j = 0; j < MT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int nk = (p.K + BK - 1) / BK;
  stage(0, 0);
  int buf = 0;
  for (int kt = 0; kt < nk; ++kt) {
    asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
    const char* la = lds + buf * STAGE_BYTES;
    const char* lb = la + A_BYTES;
    bf16x8 xf[MT];
#pragma unroll
    for (int i = 0; i < MT; ++i) xf[i] = *reinterpret_cast<const bf16x8*>(la + fx[i]);
    if (kt + 1 < nk) stage(buf ^ 1, (kt + 1) * BK);
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const bf16x8 wf = *reinterpret_cast<const bf16x8*>(lb + fw[nt]);
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) acc[nt][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf, xf[mt], acc[nt][mt], 0, 0, 0);
    }
    __builtin_amdgcn_s_setprio(0);
    buf ^= 1;
  }
  if (p.no_epi) {
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
      for (int j = 0; j < MT; ++j) asm volatile("" ::"v"(acc[i][j]));
    return;
  }
  // bias + residual, straight from the accumulators: a lane holds 4 consecutive columns of one row per 16x16 block
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    const int n = wave * 96 + nt * 16 + q4 * 4;
    f32x4 b4 = {0.f, 0.f, 0.f, 0.f};
    if (p.bias != nullptr && n + 3 < p.N) b4 = *reinterpret_cast<const f32x4*>(p.bias + n);
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const int m = m0 + mt * 16 + c16;
      if (m >= p.M || n + 3 >= p.N) continue;
      f32x4 v = acc[nt][mt] + b4;
      if (p.aux != nullptr) {
        const bf16x4 r = *reinterpret_cast<const bf16x4*>(p.aux + (int64_t)m * p.ld_aux + n);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] += bf2f(r[e]);
      }
      bf16x4 o = {f2bf(v[0]), f2bf(v[1]), f2bf(v[2]), f2bf(v[3])};
      *reinterpret_cast<bf16x4*>((__bf16*)p.out0 + (int64_t)m * p.ldc + n) = o;
    }
  }
}

int launch_rowtile(GemmNtArgs a, hipStream_t s) {
  if (a.N != 768 || (a.K & 31)) MVPTR_FAIL(MVPTR_BAD_ARG, "gemm_nt: MVPTR_GEMM_CFG=n768 needs N = 768 and K %% 32 == 0");
  constexpr int LDS_BYTES = 2 * (128 + 768) * 32 * 2;
  hipError_t e = hipFuncSetAttribute((const void*)gemm_nt_rowtile_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
  if (e != hipSuccess) MVPTR_FAIL(MVPTR_HIP_ERROR, "gemm_nt: set LDS size: %s", hipGetErrorString(e));
  hipLaunchKernelGGL(gemm_nt_rowtile_kernel, dim3((a.M + 127) / 128), dim3(512), LDS_BYTES, s, a);
  MVPTR_CHECK_LAUNCH("gemm_nt");
  return MVPTR_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// EXPERIMENT (diagnostic build only, MVPTR_GEMM_CFG=p): persistent ring form ("P", round 4) of the 256 x 256 tile for
// the encoder-layer GEMMs (N % 256 == 0, K % 32 == 0).  Built to close the gap to hipBLASLt's plain kernels
// (tools/blas_table.py: 1.05-1.8 x faster than gemm_nt_kernel on every GEMM shape of the step, cold operands).
// MEASURED AND NOT FASTER — kept for the record and the next attempt (profiles/r04_experiments.txt):
//   * first form: the BK 64 double-buffered loop of gemm_nt_kernel made persistent (next tile's two stages requested
//     in front of the stores, epilogue straight from the registers, no workgroup turnover): the SAME times as
//     gemm_nt_kernel on all 24 shape x row-count cases (sum 4 220 vs 4 164 us) — the serial cost per tile is neither
//     the turnover nor the LDS restage nor the store drain;
//   * this form: BK 32, FOUR 32-KiB stages, three in flight (96 KiB against 64), LDS-DMA from inline asm with
//     hand-counted waits (a true ring: hipcc drains the builtin form with vmcnt(0), so the BK 32 rings of rounds 1-3
//     never had more than one stage in flight), the four LDS-DMA instructions of a step spread between the MFMA
//     groups: 8 % SLOWER (4 520 us) — the loop is not bound by bytes in flight either;
//   * stores dropped at the descriptor (instructions still issued): -22 us of 189 (Q/K/V, M = 37 748), -88 of 302
//     (FFN1 + GELU); loop-only build of gemm_nt_kernel: 139 us = what hipBLASLt needs for the whole GEMM.  The cost is
//     the STORE ISSUE of the epilogue (~70 cycles per 1-KiB store instruction per CU = 14 B/clk: 4.7 us per 128-KiB
//     tile) serialised with the wave's own MFMAs — overlapping it needs the two waves of a SIMD half a tile apart,
//     which a shared operand ring does not allow;
//   * sc1 (write-through) stores: FFN1 + GELU 302 -> 237 us in this kernel (its 8-byte gelu' stores are partial
//     lines), nothing or worse elsewhere.
// What it does:
//   * the ring never drains between tiles: a workgroup keeps its CU and walks its tiles as one continuous stream of
//     stages; the stages of tile i+1 that are requested during tile i's last steps sit IN FRONT of tile i's stores
//     in the wave's in-order vector-memory queue, so the first three steps of a tile wait with vmcnt(2 * LPS + S)
//     (S = the stores of the previous epilogue, a compile-time constant: buffer stores with a per-tile descriptor
//     drop the rows past M instead of branching around them) and the stores have three steps to drain;
//   * the epilogue works on the accumulators where they are: v_permlane16_swap_b32 trades the odd 16-lane rows of
//     one 16 x 16 block with the even rows of its neighbour, which leaves every lane 8 CONSECUTIVE output columns
//     (16 bytes of bf16) — no LDS round trip, no barrier, 16-byte aux loads and stores (16 rows x 64 B per wave
//     instruction), bias / residual / gelu / gelu' arithmetic unchanged.
// Tile order = the same XCD-aware order as above (virtual block id = blockIdx.x + i * gridDim.x keeps a workgroup on
// the logical tiles of its own XCD when the grid is a multiple of 8).  grid = tiles / ceil(tiles / CUs): every
// workgroup gets the same number of tiles (+-1).
template <int EPI>
__device__ __forceinline__ constexpr int ntp_stores() {
  // vector-memory instructions a wave issues in EVERY epilogue (a lower bound is what the waits need; the optional
  // bias-gradient atomics of EPI_GELU_BWD come after the stores and are not counted)
  return EPI == MVPTR_EPI_BIAS_GELU ? 32 : 16;
}

template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    static_for<I + 1, N>(f);
  }
}

// DEFER (MVPTR_GEMM_CFG=pd): the finished tile's bf16 output stays in 64 registers per lane and ONE store goes out behind each
// of the first 16 K-steps of the next tile (a CU sustains ~70 cycles per store instruction: 128 of them in a burst hold the
// address path — and the LDS-DMA behind them — for 4.5 us; one per step is noise).  The BK 32 ring leaves the registers for it
// (182 without).  EPI_BIAS_GELU defers gelu(u) and stores the 8-bit gelu' stash at once.
template <int EPI, bool DEFER>
__global__ __launch_bounds__(512, 2) void gemm_ntp_kernel(GemmNtArgs p) {
  constexpr int OP_BYTES = 256 * 64, STAGE_BYTES = 2 * OP_BYTES, NSTAGE = 4;
  constexpr int MT = 8, LPS = 4, S = DEFER ? (EPI == MVPTR_EPI_BIAS_GELU ? 16 : 0) : ntp_stores<EPI>();   // stores issued AT the tile end
  constexpr bool kBias = (EPI == MVPTR_EPI_BIAS || EPI == MVPTR_EPI_BIAS_GELU || EPI == MVPTR_EPI_BIAS_RESID);
  constexpr bool kAux = (EPI == MVPTR_EPI_BIAS_RESID || EPI == MVPTR_EPI_GELU_BWD || EPI == MVPTR_EPI_ADD);
  constexpr int AUXW = (EPI == MVPTR_EPI_GELU_BWD) ? 2 : 4;   // dwords of aux per lane and (row block, column pair)
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nwg = p.tiles_m * p.tiles_n;
  const int G = gridDim.x;
  const int Mv = rows_clamped(p.M, p.rows_dev);
  const uint32_t lds0 = lds_addr(lds);

  // logical tile -> (m0, n0): the order of gemm_nt_kernel
  auto tile_m0n0 = [&](int vb, int& m0, int& n0) {
    const int t = xcd_remap(vb, nwg);
    const int chunk_full = p.tiles_m * p.group_n;
    const int chunk = t / chunk_full;
    const int cn0 = chunk * p.group_n;
    const int cn = min(p.group_n, p.tiles_n - cn0);
    const int tc = t - chunk * chunk_full;
    const int gsz = p.group_m * cn;
    const int grp = tc / gsz;
    const int first_m = grp * p.group_m;
    const int gm = min(p.group_m, p.tiles_m - first_m);
    const int in_g = tc - grp * gsz;
    m0 = __builtin_amdgcn_readfirstlane((first_m + in_g % gm) * 256);
    n0 = __builtin_amdgcn_readfirstlane((cn0 + in_g / gm) * 256);
  };
  auto operand_rsrc = [&](int m0, int n0, u32x4& rsA, u32x4& rsB) {
    const int rows_a = min(256, Mv - m0);
    rsA = make_rsrc_words(p.A + (int64_t)m0 * p.lda, (uint32_t)(((int64_t)(rows_a - 1) * p.lda + p.K) * 2));
    rsB = make_rsrc_words(p.B + (int64_t)n0 * p.ldb, (uint32_t)(((int64_t)255 * p.ldb + p.K) * 2));
  };

  // staging: a wave instruction fills 16 LDS rows of 64 bytes (1 KiB, lane-linear: lane -> row lane / 4, 16-byte
  // position lane % 4); 2 per wave for A, 2 for B.  Instruction i of a wave covers rows (i * 8 + wave) * 16 + lane / 4:
  // the swizzle term (LUT of (row >> 2) & 3 = (lane >> 4) & 3) does not depend on i or the wave, so one per-lane base
  // per operand + a uniform row step is all the addressing the loop keeps in registers
  uint32_t offA0, offB0;
  {
    const int row = wave * 16 + (lane >> 2);
    const int c = (lane & 3) ^ swz_row(row, 4);
    offA0 = (uint32_t)(row * p.lda * 2 + c * 16);
    offB0 = (uint32_t)(row * p.ldb * 2 + c * 16);
  }
  const uint32_t stepA = (uint32_t)(128 * p.lda * 2), stepB = (uint32_t)(128 * p.ldb * 2);
  // piece j (0..3) of a stage: A instructions 0, 1, then B instructions 0, 1
  auto stage_piece = [&](int buf, const u32x4& rsA, const u32x4& rsB, int k0, int j) {
    const uint32_t la = lds0 + (uint32_t)(buf * STAGE_BYTES + wave * 1024);
    if (j < 2) lds_dma16_add(rsA, offA0, (uint32_t)j * stepA + (uint32_t)k0 * 2, la + j * 8192);
    else lds_dma16_add(rsB, offB0, (uint32_t)(j - 2) * stepB + (uint32_t)k0 * 2, la + OP_BYTES + (j - 2) * 8192);
  };

  const int wm = wave >> 2, wn = wave & 3;
  const int c16 = lane & 15, q4 = lane >> 4;
  // fragment reads: row r of an operand tile sits at r * 64 bytes, 16-byte chunk q4 ^ LUT[(r >> 2) & 3]; the rows a
  // lane reads (block * 16 + c16) share the swizzle term, so block i is an immediate offset of i * 1024 bytes
  uint32_t fx0, fw0;
  {
    const int rx = wm * 128 + c16, rw = wn * 64 + c16;
    fx0 = rx * 64 + ((q4 ^ swz_row(rx, 4)) << 4);
    fw0 = OP_BYTES + rw * 64 + ((q4 ^ swz_row(rw, 4)) << 4);
  }

  f32x4 acc[4][MT];  // [nt][mt]
  const int nk = p.K >> 5;      // >= 8 (launch rule)

  int my = blockIdx.x;
  if (my >= nwg) return;
  const int ntiles = (nwg - my + G - 1) / G;
  const int total = ntiles * nk;     // stages this workgroup streams
  // issue cursor: the tile / K offset of the next stage to request, three stages ahead of the compute cursor
  int iss_tile = my, iss_k = 0, iss_g = 0, iss_buf = 0;
  int m0, n0;
  tile_m0n0(my, m0, n0);
  u32x4 rsA, rsB;
  operand_rsrc(m0, n0, rsA, rsB);
  auto advance_issue = [&]() {
    ++iss_g;
    iss_buf = (iss_buf + 1) & (NSTAGE - 1);
    iss_k += 32;
    if (iss_k == p.K) {
      iss_k = 0;
      iss_tile += G;
      if (iss_tile < nwg) {
        int im0, in0;
        tile_m0n0(iss_tile, im0, in0);
        operand_rsrc(im0, in0, rsA, rsB);
      }
    }
  };
#pragma unroll
  for (int st = 0; st < NSTAGE - 1; ++st) {
    if (iss_g < total) {
#pragma unroll
      for (int j = 0; j < 4; ++j) stage_piece(iss_buf, rsA, rsB, iss_k, j);
      advance_issue();
    }
  }
#define NTP_WAIT_BARRIER(n) asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(n) : "memory")

  int g = 0, buf = 0;
  // one K-step on the current ring buffer: 12 fragment reads, four groups of eight MFMAs with one LDS-DMA instruction of
  // stage g + 3 behind each (into the buffer step g - 1 read: every wave has passed this step's barrier, so it is free)
  auto kbody = [&]() {
    const char* base = lds + buf * STAGE_BYTES;
    bf16x8 xf[MT], wf[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) wf[i] = *reinterpret_cast<const bf16x8*>(base + fw0 + i * 1024);
#pragma unroll
    for (int i = 0; i < MT; ++i) xf[i] = *reinterpret_cast<const bf16x8*>(base + fx0 + i * 1024);
    const bool do_issue = iss_g < total;
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int grp = 0; grp < 4; ++grp) {
#pragma unroll
      for (int mt = 2 * grp; mt < 2 * grp + 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
          acc[nt][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[nt], xf[mt], acc[nt][mt], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      if (do_issue) stage_piece(iss_buf, rsA, rsB, iss_k, grp);
      __builtin_amdgcn_sched_barrier(0);
    }
    __builtin_amdgcn_s_setprio(0);
    if (do_issue) advance_issue();
    ++g;
    buf = (buf + 1) & (NSTAGE - 1);
  };
  // deferred output of the previous tile (DEFER): 16 x 16 bytes per lane, its descriptor, the lane's byte offset in a tile
  u32x4 pend[DEFER ? MT : 1][2];
  __amdgpu_buffer_rsrc_t rsP = make_rsrc_uniform(p.out0, 0u);
  uint32_t pend_off = 0;
  auto store_pending = [&](auto k_tag) {
    constexpr int k = decltype(k_tag)::value;
    const uint32_t vo = pend_off + (uint32_t)(((k >> 1) * 16 * p.ldc + (k & 1) * 32) * 2);
    __builtin_amdgcn_raw_buffer_store_b128(pend[DEFER ? (k >> 1) : 0][k & 1], rsP, vo, 0, 0);
  };
  for (int ti = 0; ti < ntiles; ++ti) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < MT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    int kt = 0;
    if (DEFER && ti > 0) {
      // The queue behind stage g's LDS-DMA holds the two younger stages (8) and the pending stores of the last three steps
      // (one per step while steps 0..15 run), plus — in the first three steps — whatever the previous tile end stored at once.
      static_for<0, 19>([&](auto kt_tag) {
        constexpr int KT = decltype(kt_tag)::value;
        constexpr int ST = KT <= 16 ? (KT < 3 ? KT : 3) : (KT == 17 ? 2 : 1);
        NTP_WAIT_BARRIER(2 * LPS + ST + (KT < 3 ? S : 0));
        kbody();
        if constexpr (KT < 16) store_pending(kt_tag);
      });
      kt = 19;
    }
    for (; kt < nk; ++kt) {
      // stage g has landed once only the (up to two) younger stages — and, in the first three steps of a tile that
      // is not the workgroup's first, the previous epilogue's stores, which were issued behind them — remain
      const int younger = min(NSTAGE - 2, total - 1 - g);
      if (!DEFER && ti > 0 && kt < NSTAGE - 1) NTP_WAIT_BARRIER(2 * LPS + S);
      else if (younger == 2) NTP_WAIT_BARRIER(2 * LPS);
      else if (younger == 1) NTP_WAIT_BARRIER(LPS);
      else NTP_WAIT_BARRIER(0);
      kbody();
    }
    // ------------------------------------------------------------------ epilogue, from the accumulators
    // epilogue geometry: after the lane-row swap a lane owns 8 consecutive columns of the pair's 32: block (q4 & 1) of
    // the pair, half (q4 >> 1) of the block.  Derived from an opaque copy of the lane id so that the addresses are
    // formed here and not hoisted above the K loop (where they would be spilled to scratch).
    int lane_e = lane;
    asm volatile("" : "+v"(lane_e));
    const int e16 = lane_e & 15, eq4 = lane_e >> 4;
    const int ecol = wn * 64 + (eq4 & 1) * 16 + (eq4 >> 1) * 8;   // + pr * 32
    const int erow = wm * 128 + e16;                              // + mt * 16
    const int rows_valid = min(256, Mv - m0);
    // residual / gelu' rows of the wave block: the first four row blocks are requested here, in front of the next
    // tile's stage 1; block mt + 4 after block mt has been finished (its accumulator registers are free by then)
    u32x4 auxr[kAux ? MT : 1][2];
    const bool has_aux = kAux && p.aux != nullptr;
    constexpr int xsz = (EPI == MVPTR_EPI_GELU_BWD) ? 1 : 2;
    const __amdgpu_buffer_rsrc_t rsX = make_rsrc_uniform(
        reinterpret_cast<const char*>(p.aux) + ((int64_t)m0 * p.ld_aux + n0) * xsz,
        has_aux ? (uint32_t)(((int64_t)(rows_valid - 1) * p.ld_aux + 256) * xsz) : 0u);
    const uint32_t xoff = (uint32_t)((erow * p.ld_aux + ecol) * xsz);
    auto load_aux = [&](int mt) {
      if constexpr (kAux) {
#pragma unroll
        for (int pr = 0; pr < 2; ++pr) {
          const uint32_t vo = xoff + (uint32_t)((mt * 16 * p.ld_aux + pr * 32) * xsz);
          if constexpr (AUXW == 2) {
            const u32x2 w = __builtin_amdgcn_raw_buffer_load_b64(rsX, vo, 0, 0);
            auxr[mt][pr] = u32x4{w[0], w[1], 0u, 0u};
          } else {
            auxr[mt][pr] = __builtin_amdgcn_raw_buffer_load_b128(rsX, vo, 0, 0);
          }
        }
      }
    };
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) load_aux(mt);
    // bias of this tile's columns (16 floats per lane)
    float b8[2][8];
#pragma unroll
    for (int pr = 0; pr < 2; ++pr)
#pragma unroll
      for (int e = 0; e < 8; ++e) b8[pr][e] = 0.f;
    if constexpr (kBias) {
      if (p.bias != nullptr) {
#pragma unroll
        for (int pr = 0; pr < 2; ++pr) {
          const f32x4 lo = *reinterpret_cast<const f32x4*>(p.bias + n0 + ecol + pr * 32);
          const f32x4 hi = *reinterpret_cast<const f32x4*>(p.bias + n0 + ecol + pr * 32 + 4);
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            b8[pr][e] = lo[e];
            b8[pr][4 + e] = hi[e];
          }
        }
      }
    }
    {
      const int osz0 = (EPI == MVPTR_EPI_BIAS_GELU) ? 1 : 2;      // out0 of the GELU epilogue is the 8-bit gelu' stash
#ifdef MVPTR_DIAG_BUILD
      const uint32_t keep = (p.store_mode == 1) ? 0u : 1u;     // 0: every store falls outside the descriptor and is dropped
#else
      constexpr uint32_t keep = 1u;
#endif
      const __amdgpu_buffer_rsrc_t rsO = make_rsrc_uniform(
          reinterpret_cast<char*>(p.out0) + ((int64_t)m0 * p.ldc + n0) * osz0,
          keep * (uint32_t)(((int64_t)(rows_valid - 1) * p.ldc + 256) * osz0));
      const __amdgpu_buffer_rsrc_t rsO1 = (EPI == MVPTR_EPI_BIAS_GELU)
          ? make_rsrc_uniform(reinterpret_cast<char*>(p.out1) + ((int64_t)m0 * p.ldc + n0) * 2,
                              keep * (uint32_t)(((int64_t)(rows_valid - 1) * p.ldc + 256) * 2))
          : rsO;
      const uint32_t ooff = (uint32_t)(erow * p.ldc + ecol);    // elements
      if constexpr (DEFER) {
        rsP = (EPI == MVPTR_EPI_BIAS_GELU) ? rsO1 : rsO;
        pend_off = ooff * 2;
      }
      float cs[2][8];
#pragma unroll
      for (int pr = 0; pr < 2; ++pr)
#pragma unroll
        for (int e = 0; e < 8; ++e) cs[pr][e] = 0.f;
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        if (mt >= 1 && mt + 3 < MT) load_aux(mt + 3);     // block mt - 1 is finished: its registers are free
#pragma unroll
        for (int pr = 0; pr < 2; ++pr) {
          float v[8];
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            // (scalar copies first: __builtin_bit_cast applied to a vector ELEMENT reads element 0 whatever the index,
            //  hipcc / ROCm 7.2)
            const float ea = acc[2 * pr][mt][r], eb = acc[2 * pr + 1][mt][r];
            const auto sw = __builtin_amdgcn_permlane16_swap(__builtin_bit_cast(uint32_t, ea), __builtin_bit_cast(uint32_t, eb), false, false);
            const uint32_t s0 = sw[0], s1 = sw[1];
            v[r] = __builtin_bit_cast(float, s0);
            v[4 + r] = __builtin_bit_cast(float, s1);
          }
          if constexpr (kBias) {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] += b8[pr][e];
          }
          float a[8];
          if constexpr (kAux) {
            const u32x4 aw = auxr[mt][pr];
            if constexpr (EPI == MVPTR_EPI_GELU_BWD) {
#pragma unroll
              for (int e = 0; e < 8; ++e) a[e] = dgelu_unpack(aw[e >> 2], e & 3);
            } else {
              const bf16x8 ab = __builtin_bit_cast(bf16x8, aw);
#pragma unroll
              for (int e = 0; e < 8; ++e) a[e] = bf2f(ab[e]);
            }
          }
          const uint32_t eo = ooff + (uint32_t)(mt * 16 * p.ldc + pr * 32);
          auto store8 = [&](const __amdgpu_buffer_rsrc_t& rs, const float (&x)[8]) {
            bf16x8 o;
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = f2bf(x[e]);
            if constexpr (DEFER) {
              pend[DEFER ? mt : 0][pr] = __builtin_bit_cast(u32x4, o);      // goes out during the next tile's first 16 steps
              return;
            }
#ifdef MVPTR_DIAG_BUILD
            if (p.store_mode == 2) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o), rs, eo * 2, 0, 2);
            else if (p.store_mode == 3) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o), rs, eo * 2, 0, 16);
            else if (p.store_mode == 4) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o), rs, eo * 2, 0, 17);
            else
#endif
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o), rs, eo * 2, 0, 0);
          };
          if constexpr (EPI == MVPTR_EPI_BIAS) {
            store8(rsO, v);
          } else if constexpr (EPI == MVPTR_EPI_BIAS_GELU) {
            float g[8], dg[8];
#pragma unroll
            for (int e = 0; e < 8; e += 2) {
              f32x2 a2, d2;
              gelu_pair(f32x2{v[e], v[e + 1]}, a2, d2);
              g[e] = a2.x;
              g[e + 1] = a2.y;
              dg[e] = d2.x;
              dg[e + 1] = d2.y;
            }
            const u32x2 dq = {dgelu_pack4(dg[0], dg[1], dg[2], dg[3]), dgelu_pack4(dg[4], dg[5], dg[6], dg[7])};
            __builtin_amdgcn_raw_buffer_store_b64(dq, rsO, eo, 0, 2);      // aux 2 = nt: read once, in the backward pass
            store8(rsO1, g);
          } else if constexpr (EPI == MVPTR_EPI_BIAS_RESID) {
            const uint64_t di = (uint64_t)(m0 + erow + mt * 16) * (uint64_t)p.N + (uint64_t)(n0 + ecol + pr * 32);
#pragma unroll
            for (int e = 0; e < 8; e += 2) drop_apply2(drop_resolve(p.drop), di + (uint64_t)e, v[e], v[e + 1]);
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] += a[e];
            store8(rsO, v);
          } else if constexpr (EPI == MVPTR_EPI_GELU_BWD) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
              v[e] *= a[e];
              cs[pr][e] += v[e];
            }
            store8(rsO, v);
          } else {   // MVPTR_EPI_ADD
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] += a[e];
            store8(rsO, v);
          }
        }
      }
      if constexpr (EPI == MVPTR_EPI_GELU_BWD) {
        if (p.vec_out != nullptr) {
          // bias gradient: column sums over the wave's 128 rows = over mt (above) and over the 16 lanes of a row
#pragma unroll
          for (int pr = 0; pr < 2; ++pr)
#pragma unroll
            for (int e = 0; e < 8; ++e) {
              float sum = cs[pr][e];
              sum += __shfl_xor(sum, 1);
              sum += __shfl_xor(sum, 2);
              sum += __shfl_xor(sum, 4);
              sum += __shfl_xor(sum, 8);
              if (e16 == 0) atomicAdd(p.vec_out + n0 + ecol + pr * 32 + e, sum);
            }
        }
      }
    }
    my += G;
    if (ti + 1 < ntiles) tile_m0n0(my, m0, n0);
  }
  if constexpr (DEFER) static_for<0, 16>([&](auto k_tag) { store_pending(k_tag); });      // the last tile's output
#undef NTP_WAIT_BARRIER
}


template <int EPI>
bool ntp_eligible(const GemmNtArgs& a) {
  if constexpr (!(EPI == MVPTR_EPI_BIAS || EPI == MVPTR_EPI_BIAS_GELU || EPI == MVPTR_EPI_BIAS_RESID || EPI == MVPTR_EPI_GELU_BWD ||
                  EPI == MVPTR_EPI_ADD))
    return false;
  if ((a.N & 255) || (a.K & 31) || a.K < 256 || a.splits > 1 || a.k_split_len > 0) return false;
  if (!a.vec_out_ok || (a.bias && !a.vec_bias_ok) || (a.aux && !a.vec_aux_ok)) return false;
  if ((EPI == MVPTR_EPI_BIAS_RESID || EPI == MVPTR_EPI_GELU_BWD) && !a.aux) return false;
  if (EPI == MVPTR_EPI_BIAS_RESID && (a.N & 1)) return false;
  // 32-bit buffer offsets inside a tile
  if ((int64_t)256 * a.lda * 2 >= (int64_t)0x7fffffff || (int64_t)256 * a.ldb * 2 >= (int64_t)0x7fffffff ||
      (int64_t)256 * a.ldc * 2 >= (int64_t)0x7fffffff || (int64_t)256 * a.ld_aux * 2 >= (int64_t)0x7fffffff)
    return false;
  return true;
}

template <int EPI, bool DEFER>
int launch_ntp(GemmNtArgs a, hipStream_t s) {
  constexpr int LDS_BYTES = 4 * 2 * 256 * 64;
  a.tiles_m = (a.M + 255) / 256;
  a.tiles_n = a.N / 256;
  hipError_t e = hipFuncSetAttribute((const void*)gemm_ntp_kernel<EPI, DEFER>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
  if (e != hipSuccess) MVPTR_FAIL(MVPTR_HIP_ERROR, "gemm_nt: set LDS size: %s", hipGetErrorString(e));
  const int nwg = a.tiles_m * a.tiles_n;
  const bool chunked = a.tiles_n > 4;
  a.group_m = (EPI == MVPTR_EPI_BIAS_GELU && chunked) ? 6 : GROUP_M;
  a.group_n = chunked ? ((EPI == MVPTR_EPI_GELU_BWD) ? 3 : 4) : a.tiles_n;
  if (mvptr_knobs().nt_group[0] > 0) a.group_m = mvptr_knobs().nt_group[0];
  if (mvptr_knobs().nt_group[1] > 0) a.group_n = min(mvptr_knobs().nt_group[1], a.tiles_n);
  if (mvptr_knobs().nt_group[0] > 0 && mvptr_knobs().nt_group[1] <= 0) a.group_n = a.tiles_n;
  // equal shares: tiles / ceil(tiles / CUs) workgroups (a multiple of 8 where that costs no extra round, so that a
  // workgroup's tiles stay on its XCD's part of the tile order)
  const int ncu = nt_num_cus();
  const int rounds = (nwg + ncu - 1) / ncu;
  int grid = (nwg + rounds - 1) / rounds;
  const int grid8 = (grid + 7) & ~7;
  if (grid8 <= ncu && grid8 <= nwg) grid = grid8;
  hipLaunchKernelGGL((gemm_ntp_kernel<EPI, DEFER>), dim3(grid), dim3(512), LDS_BYTES, s, a);
  MVPTR_CHECK_LAUNCH("gemm_nt");
  return MVPTR_OK;
}

template <int EPI>
int diag_dispatch(const GemmNtArgs& a, const char* env, hipStream_t s) {
  if (env[0] == 'n') {                                                  // "n768": row-owning tile experiment
    if constexpr (EPI == MVPTR_EPI_BIAS_RESID || EPI == MVPTR_EPI_BIAS || EPI == MVPTR_EPI_ADD) return launch_rowtile(a, s);
    else MVPTR_FAIL(MVPTR_BAD_ARG, "gemm_nt: MVPTR_GEMM_CFG=n768 supports the bias / residual / add epilogues only");
  }
  if (env[0] == 'p') {                                                  // "p": persistent ring experiment; "pd": deferred stores
    if constexpr (EPI == MVPTR_EPI_BIAS || EPI == MVPTR_EPI_BIAS_GELU || EPI == MVPTR_EPI_BIAS_RESID || EPI == MVPTR_EPI_GELU_BWD ||
                  EPI == MVPTR_EPI_ADD) {
      if (ntp_eligible<EPI>(a)) return (env[1] == 'd' && a.K >= 32 * 24) ? launch_ntp<EPI, true>(a, s) : launch_ntp<EPI, false>(a, s);
    }
  }
  return MVPTR_DIAG_NOT_HANDLED;
}
}  // namespace

int mvptr_diag_gemm_nt(int epilogue, const GemmNtArgs& a, const char* cfg, hipStream_t s) {
  switch (epilogue) {
    case MVPTR_EPI_BIAS: return diag_dispatch<MVPTR_EPI_BIAS>(a, cfg, s);
    case MVPTR_EPI_BIAS_GELU: return diag_dispatch<MVPTR_EPI_BIAS_GELU>(a, cfg, s);
    case MVPTR_EPI_BIAS_RESID: return diag_dispatch<MVPTR_EPI_BIAS_RESID>(a, cfg, s);
    case MVPTR_EPI_GELU_BWD: return diag_dispatch<MVPTR_EPI_GELU_BWD>(a, cfg, s);
    case MVPTR_EPI_ADD: return diag_dispatch<MVPTR_EPI_ADD>(a, cfg, s);
    default: return MVPTR_DIAG_NOT_HANDLED;
  }
}
#endif  // MVPTR_DIAG_BUILD
