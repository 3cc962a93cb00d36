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
#include "common.h"
#include <stdlib.h>
#include <string.h>
#include <type_traits>

namespace {

constexpr int ST_LD = 68;  // f32 row stride of the epilogue staging block
constexpr int GROUP_M = 4;
// BK = 64: 3 x 48 KiB ring, one workgroup per CU.  BK = 32: 3 x 24 KiB ring, two workgroups per
// CU, so one workgroup's epilogue (stores) overlaps the other's MFMA main loop.
template <int BK, int STAGES, int WM, int WN, int MT_>
struct Cfg {
  static constexpr int BM = WM * MT_ * 16;         // workgroup tile rows
  static constexpr int BN = WN * 64;               // workgroup tile columns (each wave owns 64)
  static constexpr int A_BYTES = BM * BK * 2;
  static constexpr int B_BYTES = BN * BK * 2;
  static constexpr int STAGE_BYTES = A_BYTES + B_BYTES;
  static constexpr int LDS_BYTES = STAGES * STAGE_BYTES;
  static constexpr int ROW_B = BK * 2;             // bytes per LDS row
  static constexpr int CHUNKS = BK / 8;            // 16-byte chunks per row
  static constexpr int ROWS_PER_INSTR = 1024 / ROW_B;
  static constexpr int NWAVES = WM * WN;           // waves as WM (M) x WN (N)
  static constexpr int MT = MT_;                   // 16-row MFMA tiles per wave (4: 64 rows, 8: 128 rows)
  static constexpr int WG_PER_CU = (STAGES * (BM + BN) * BK * 2 <= 80 * 1024) ? 2 : 1;
  static constexpr int NA = BM / ROWS_PER_INSTR / NWAVES;  // A staging instructions per wave
  static constexpr int NB = BN / ROWS_PER_INSTR / NWAVES;  // B staging instructions per wave
  static constexpr int KS = BK / 32;               // MFMA k-substeps per stage
};
// chunk swizzles that make the 16x16x32 ds_read_b128 fragment reads conflict free
__device__ __forceinline__ int swz_row(int row, int chunks) {
  return chunks == 8 ? ((row >> 1) & 7) : ((0x78 >> (((row >> 2) & 3) * 2)) & 3);  // LUT {0,2,3,1}
}

struct GemmNtArgs {
  const __bf16* A;
  const __bf16* B;
  int64_t lda, ldb;
  int M, N, K;
  const float* bias;
  const __bf16* aux;
  int64_t ld_aux;
  void* out0;
  void* out1;
  int64_t ldc;
  float* vec_out;
  DropDev drop;
  int tiles_m, tiles_n;
  int group_m, group_n;  // tile order: column tiles in chunks of group_n, inside a chunk group_m row tiles x the chunk's columns, row tile fastest
  int vec_out_ok;   // 16-byte stores allowed on out0/out1
  int vec_aux_ok;   // 16-byte loads allowed on aux
  int vec_bias_ok;  // 16-byte loads allowed on bias
  int delay_cycles;  // experiment (MVPTR_GEMM_DELAY): start delay of the second resident workgroups
  int delay_lo, delay_hi;
  int exp_flags;  // experiments (MVPTR_NT_EXP): bit 0 = persistent kernel without next-tile prefetch
  unsigned long long* stamps;  // diagnostic build only (MVPTR_GEMM_STAMPS): per-workgroup cycle sums
  // fused vocabulary decoder + cross entropy (mvptr_decoder_ce_fwd / _bwd)
  const int64_t* labels;  // [M], < 0 or >= N: row not scored
  const float* lse;       // [M] row log-sum-exp (backward)
  const float* scale;     // [1] d(loss)/d(row loss) (backward)
  float* part;            // [M, part_ld, 2] per-64-column (max, sum exp) partials (forward)
  float* lab_logit;       // [M] logit at the label (forward)
  int part_ld;
  int n_store;            // columns written by EPI_CE_BWD (N rounded up to the operand padding)
};
// library-internal epilogues of the fused decoder + cross-entropy entry points
constexpr int EPI_CE_PART = 7;  // per row and 64-column wave strip: (max, sum exp(v - max)) of v = acc + bias; logit at the label
constexpr int EPI_CE_BWD = 8;   // out0(bf16) = (exp(v - lse[m]) - [n == label[m]]) * scale, 0 for unscored rows / pad columns

extern __shared__ __attribute__((aligned(1024))) char lds[];

// Epilogue of one output tile: the wave's accumulators (MFMA layout: a lane holds 4 consecutive
// columns of one row per 16x16 block) are restaged through the wave's private LDS area `st` in
// CHUNK-row pieces and finished in row-chunk form (8 consecutive columns per lane): 16-byte bias /
// residual loads, 16-byte coalesced stores.  CHUNK 32: 32 x ST_LD floats per wave (inside the
// operand ring); CHUNK 16: 16 x 64 floats, XOR-swizzled (4 KiB per wave, beside the ring).
template <int EPI, int MT, int CHUNK>
__device__ __forceinline__ void nt_epilogue(const GemmNtArgs& p, f32x4 (&acc)[4][MT], float* st, int m0, int n0,
                                            int wm, int wn, int lane) {
  constexpr int WROWS = MT * 16;
  const int c16 = lane & 15, q4 = lane >> 4;

  const int ch = lane & 7, rsub = lane >> 3;
  const int n = n0 + wn * 64 + ch * 8;
  const bool nfull = (n + 7 < p.N);
  float b8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (EPI != MVPTR_EPI_GELU_BWD && EPI != MVPTR_EPI_ADD && p.bias != nullptr && n < p.N) {
    if (nfull && p.vec_bias_ok) {
      const f32x4 b0 = *reinterpret_cast<const f32x4*>(p.bias + n);
      const f32x4 b1 = *reinterpret_cast<const f32x4*>(p.bias + n + 4);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        b8[e] = b0[e];
        b8[4 + e] = b1[e];
      }
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e)
        if (n + e < p.N) b8[e] = p.bias[n + e];
    }
  }
  float cs[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};

  auto store_bf8 = [&](void* base, int m, const float v[8]) {
    __bf16* op = (__bf16*)base + (int64_t)m * p.ldc + n;
    if (nfull && p.vec_out_ok) {
      bf16x8 o;
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = f2bf(v[e]);
      if (p.exp_flags & 256) __builtin_nontemporal_store(o, reinterpret_cast<bf16x8*>(op));  // A/B knob (MVPTR_NT_EXP bit 8)
      else *reinterpret_cast<bf16x8*>(op) = o;
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e)
        if (n + e < p.N) op[e] = f2bf(v[e]);
    }
  };

  constexpr bool kNeedsAux = (EPI == MVPTR_EPI_BIAS_RESID || EPI == MVPTR_EPI_GELU_BWD || EPI == MVPTR_EPI_ADD);
  const bool has_aux = kNeedsAux && p.aux != nullptr;

#pragma unroll
  for (int it = 0; it < MT * 2; ++it) {
    // 32-row chunk ck = it >> 2 of the wave's block goes through the staging area; its residual /
    // pre-activation rows are requested together so the chunk pays one memory latency, not four
    bf16x8 auxv[4];
    if constexpr (CHUNK == 32) {
      if ((it & 3) == 0) {
        const int ck = it >> 2;
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
          for (int mh = 0; mh < 2; ++mh)
            *reinterpret_cast<f32x4*>(st + (mh * 16 + c16) * ST_LD + nt * 16 + q4 * 4) = acc[nt][2 * ck + mh];
      }
    } else {
      // 16-row chunks, unpadded 64-float rows, 16-byte chunk index XOR row: conflict free for the
      // 4x4-block writes and for the row reads below
      if ((it & 1) == 0) {
        const int ck = it >> 1;
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
          *reinterpret_cast<f32x4*>(st + c16 * 64 + (((nt * 4 + q4) ^ c16) << 2)) = acc[nt][ck];
      }
    }
    if (kNeedsAux && (it & 3) == 0) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int mj = m0 + wm * WROWS + (it + j) * 8 + rsub;
        bf16x8 x;
#pragma unroll
        for (int e = 0; e < 8; ++e) x[e] = f2bf(0.f);
        if (has_aux && mj < p.M && n < p.N) {
          const __bf16* ap = p.aux + (int64_t)mj * p.ld_aux + n;
          if (nfull && p.vec_aux_ok) {
            x = *reinterpret_cast<const bf16x8*>(ap);
          } else {
#pragma unroll
            for (int e = 0; e < 8; ++e)
              if (n + e < p.N) x[e] = ap[e];
          }
        }
        auxv[j] = x;
      }
    }
    const int row = it * 8 + rsub;
    const int lrow = row & (CHUNK - 1);
    const int m = m0 + wm * WROWS + row;
    f32x4 v0, v1;
    if constexpr (CHUNK == 32) {
      v0 = *reinterpret_cast<const f32x4*>(st + lrow * ST_LD + ch * 8);
      v1 = *reinterpret_cast<const f32x4*>(st + lrow * ST_LD + ch * 8 + 4);
    } else {
      v0 = *reinterpret_cast<const f32x4*>(st + lrow * 64 + (((2 * ch) ^ lrow) << 2));
      v1 = *reinterpret_cast<const f32x4*>(st + lrow * 64 + (((2 * ch + 1) ^ lrow) << 2));
    }
    if constexpr (EPI == EPI_CE_PART) {
      // online log-sum-exp over this lane's 8 columns, then over the 8 lanes that share the row
      // (lane bits 0-2): every lane takes part in the shuffles, masked columns count as -inf
      float mx = -1e30f, sm = 0.f, u[8];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        u[e] = v0[e] + b8[e];
        u[4 + e] = v1[e] + b8[4 + e];
      }
#pragma unroll
      for (int e = 0; e < 8; ++e)
        if (n + e < p.N) mx = fmaxf(mx, u[e]);
#pragma unroll
      for (int e = 0; e < 8; ++e)
        if (n + e < p.N) sm += __expf(u[e] - mx);
#pragma unroll
      for (int o = 1; o < 8; o <<= 1) {
        const float m2 = __shfl_xor(mx, o), s2 = __shfl_xor(sm, o);
        const float mm = fmaxf(mx, m2);
        sm = sm * __expf(mx - mm) + s2 * __expf(m2 - mm);
        mx = mm;
      }
      if (m < p.M) {
        if (ch == 0 && (n0 >> 6) + wn < p.part_ld) {  // strips past the last column do not exist
          float* pp = p.part + ((int64_t)m * p.part_ld + ((n0 >> 6) + wn)) * 2;
          pp[0] = mx;
          pp[1] = sm;
        }
        const int64_t lab = p.labels[m];
        if (lab >= n && lab < n + 8 && lab < p.N) p.lab_logit[m] = u[(int)(lab - n)];
      }
      continue;
    }
    if (m >= p.M || n >= (EPI == EPI_CE_BWD ? p.n_store : p.N)) continue;
    float v[8], a[8];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      v[e] = v0[e] + b8[e];
      v[4 + e] = v1[e] + b8[4 + e];
    }
    if constexpr (EPI == EPI_CE_BWD) {
      const int64_t lab = p.labels[m];
      const bool scored = lab >= 0 && lab < p.N;
      const float sc = scored ? p.scale[0] : 0.f, lse = p.lse[m];
      __bf16* op = (__bf16*)p.out0 + (int64_t)m * p.ldc + n;
      bf16x8 o;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float g = (n + e < p.N && scored) ? (__expf(v[e] - lse) - ((int64_t)(n + e) == lab ? 1.f : 0.f)) * sc : 0.f;
        o[e] = f2bf(g);
      }
      if (n + 7 < p.n_store && p.vec_out_ok) {
        *reinterpret_cast<bf16x8*>(op) = o;
      } else {
#pragma unroll
        for (int e = 0; e < 8; ++e)
          if (n + e < p.n_store) op[e] = o[e];
      }
      continue;
    }
    if (kNeedsAux) {
#pragma unroll
      for (int e = 0; e < 8; ++e) a[e] = bf2f(auxv[it & 3][e]);
    }
    if (EPI == MVPTR_EPI_BIAS) {
      store_bf8(p.out0, m, v);
    } else if (EPI == MVPTR_EPI_BIAS_GELU) {
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
      store_bf8(p.out0, m, dg);
      store_bf8(p.out1, m, g);
    } else if (EPI == MVPTR_EPI_BIAS_RESID) {
      if ((p.N & 1) == 0) {  // (m*N + n) even: lanes own whole hash pairs
#pragma unroll
        for (int e = 0; e < 8; e += 2)
          drop_apply2(p.drop, (uint64_t)m * (uint64_t)p.N + (uint64_t)(n + e), v[e], v[e + 1]);
      } else {
#pragma unroll
        for (int e = 0; e < 8; ++e)
          v[e] = drop_apply(p.drop, (uint64_t)m * (uint64_t)p.N + (uint64_t)(n + e), v[e]);
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] += a[e];
      store_bf8(p.out0, m, v);
    } else if (EPI == MVPTR_EPI_GELU_BWD) {
      // aux = gelu'(u) saved by the forward epilogue; rows / columns outside the problem have
      // acc = 0 (zero-filled operand rows), so the column sums need no guard
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        v[e] *= a[e];
        cs[e] += v[e];
      }
      store_bf8(p.out0, m, v);
    } else if (EPI == MVPTR_EPI_ADD) {
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] += a[e];
      store_bf8(p.out0, m, v);
    } else if (EPI == MVPTR_EPI_F32) {
      float* op = (float*)p.out0 + (int64_t)m * p.ldc + n;
      if (nfull && p.vec_out_ok) {
        *reinterpret_cast<f32x4*>(op) = f32x4{v[0], v[1], v[2], v[3]};
        *reinterpret_cast<f32x4*>(op + 4) = f32x4{v[4], v[5], v[6], v[7]};
      } else {
#pragma unroll
        for (int e = 0; e < 8; ++e)
          if (n + e < p.N) op[e] = v[e];
      }
    } else if (EPI == MVPTR_EPI_BIAS_TANH) {
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = tanhf(v[e]);
      store_bf8(p.out0, m, v);
    }
  }
  if (EPI == MVPTR_EPI_GELU_BWD && p.vec_out != nullptr) {
    // sum the 8 row-lanes (lane>>3) that share a column chunk, then one atomic per column
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      float s = cs[e];
      s += __shfl_xor(s, 8);
      s += __shfl_xor(s, 16);
      s += __shfl_xor(s, 32);
      if (rsub == 0 && n + e < p.N) atomicAdd(p.vec_out + n + e, s);
    }
  }
}

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
  const int nwg = p.tiles_m * p.tiles_n;
#ifdef MVPTR_TIMELINE_BUILD
  // diagnostic: wall-clock (100 MHz s_memrealtime) start / loop-end / end of every workgroup
  unsigned long long tl_start, tl_loop, tl_end;
  asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tl_start)::"memory");
#endif
  if (p.delay_cycles > 0) {
    // experiment: hi >= 0: workgroups lo <= id < hi start `cycles` late; hi < 0: the first `lo`
    // workgroups start (id / 8 % P) / P * cycles late, P = -hi (phases spread inside each XCD)
    long long d = 0;
    if (p.delay_hi >= 0) {
      if ((int)blockIdx.x >= p.delay_lo && (int)blockIdx.x < p.delay_hi) d = p.delay_cycles;
    } else if (p.delay_hi == -1000) {
      // phases spread BETWEEN the 8 XCD groups (blockIdx % 8), equal inside a group: the workgroups that
      // share an L2 keep sweeping K together while the groups' output bursts come at different times
      if ((int)blockIdx.x < p.delay_lo) d = (long long)p.delay_cycles * ((int)blockIdx.x & 7) / 8;
    } else if ((int)blockIdx.x < p.delay_lo) {
      const int P = -p.delay_hi;
      d = (long long)p.delay_cycles * (((int)blockIdx.x >> 3) % P) / P;
    }
    const long long t0 = __builtin_readcyclecounter();
    while (__builtin_readcyclecounter() - t0 < d) __builtin_amdgcn_s_sleep(16);
  }
  // MVPTR_NT_EXP bit 14 (A/B knob): no XCD remap — consecutive logical tiles go to consecutive XCDs
  const int t = (p.exp_flags & 16384) ? (int)blockIdx.x : xcd_remap(blockIdx.x, nwg);
  // order of the logical tiles (an XCD owns a contiguous run of them): column tiles in chunks of
  // group_n; inside a chunk, groups of group_m row tiles x the chunk's columns, row tile fastest.
  // A chunk narrower than the matrix keeps that part of B in the XCD's L2 while its rows stream by.
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
  const int tm = first_m + in_g % gm;
  const int tn = cn0 + in_g / gm;
  const int m0 = tm * BM, n0 = tn * BN;
  const int rows_a = min(BM, p.M - m0);
  const int rows_b = min(BN, p.N - n0);

  const __amdgpu_buffer_rsrc_t rsA =
      make_rsrc(p.A + (int64_t)m0 * p.lda, (uint32_t)(((int64_t)(rows_a - 1) * p.lda + p.K) * 2));
  const __amdgpu_buffer_rsrc_t rsB =
      make_rsrc(p.B + (int64_t)n0 * p.ldb, (uint32_t)(((int64_t)(rows_b - 1) * p.ldb + p.K) * 2));

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
      const uint32_t va = (k0 + kcA[i] < p.K) ? offA[i] + (uint32_t)k0 * 2 : MVPTR_OOB;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, LDS_PTR(la + (i * NWAVES + wave) * 1024), 16, va, 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      const uint32_t vb = (k0 + kcB[i] < p.K) ? offB[i] + (uint32_t)k0 * 2 : MVPTR_OOB;
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

  const int nk = (p.K + BK - 1) / BK;
  constexpr int LPS = NA + NB;  // loads per stage per thread
  if constexpr (SCHED == 2) {
    // Staggered halves: the MFMA pipe of a SIMD is shared by wave i and wave i + NWAVES/2.  If both
    // run "issue loads, then MFMAs" in lockstep, the pipe idles while every wave issues its LDS-DMA
    // and fragment reads (~400 cycles per K-step measured with s_memtime) and then serialises both
    // waves' MFMA bursts.  The second half of the waves therefore runs one burst behind: after each
    // barrier it first issues the MFMAs of the PREVIOUS K-step (fragments kept in registers across
    // the barrier) while the first half issues its loads, then swaps roles.
    static_assert(KS == 1, "stagger schedule is written for BK = 32");
#pragma unroll
    for (int s = 0; s < STAGES - 1; ++s)
      if (s < nk) stage(s, s * BK);
    const bool late = __builtin_amdgcn_readfirstlane(wave) >= NWAVES / 2;
    bf16x8 xf[MT], wf[4];
    auto mma = [&]() {
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int nt = 0; nt < 4; ++nt)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
          acc[nt][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[nt], xf[mt], acc[nt][mt], 0, 0, 0);
      __builtin_amdgcn_s_setprio(0);
    };
    int buf = 0;
    for (int kt = 0; kt < nk; ++kt) {
      const int younger = min(STAGES - 2, nk - 1 - kt);
      if (younger >= 3)
        asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(3 * LPS) : "memory");
      else if (younger == 2)
        asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(2 * LPS) : "memory");
      else if (younger == 1)
        asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(LPS) : "memory");
      else
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
      const char* la = lds + buf * STAGE_BYTES;
      const char* lb = la + A_BYTES;
      if (late && kt > 0) mma();  // K-step kt-1 of the late half, beside the early half's loads
#pragma unroll
      for (int i = 0; i < 4; ++i) wf[i] = *reinterpret_cast<const bf16x8*>(lb + fw[i][0]);
#pragma unroll
      for (int i = 0; i < MT; ++i) xf[i] = *reinterpret_cast<const bf16x8*>(la + fx[i][0]);
      if (kt + STAGES - 1 < nk) {
        int nb = buf + STAGES - 1;
        if (nb >= STAGES) nb -= STAGES;
        stage(nb, (kt + STAGES - 1) * BK);
      }
      if (!late) mma();
      buf = (buf + 1 == STAGES) ? 0 : buf + 1;
    }
    if (late) mma();
  } else {
  // prologue: STAGES-1 stages in flight
#pragma unroll
  for (int s = 0; s < STAGES - 1; ++s)
    if (s < nk) stage(s, s * BK);
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
    if (kt + STAGES - 1 < nk) {
      int nb = buf + STAGES - 1;
      if (nb >= STAGES) nb -= STAGES;
      stage(nb, (kt + STAGES - 1) * BK);
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
#pragma unroll
        for (int i = 0; i < 4; ++i) wf[i] = *reinterpret_cast<const bf16x8*>(lb + fw[i][ks]);
#pragma unroll
        for (int i = 0; i < MT; ++i) xf[i] = *reinterpret_cast<const bf16x8*>(la + fx[i][ks]);
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
  }

  // ------------------------------------------------------------------ epilogue
#ifdef MVPTR_TIMELINE_BUILD
  asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tl_loop)::"memory");
#endif
  __syncthreads();  // every wave is done with the operand ring
  nt_epilogue<EPI, MT, 32>(p, acc, reinterpret_cast<float*>(lds) + wave * (32 * ST_LD), m0, n0, wm, wn, lane);
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

// Persistent form of the 256x256 / BK 64 configuration: one workgroup per CU walks over its tiles.
// The 2 x 64 KiB operand ring and a separate 32-KiB epilogue staging area fill the 160-KiB LDS, so
// after the last K-step of a tile both ring stages are free: the LDS-DMA loads of the NEXT tile's
// first two K-steps are issued before the epilogue starts and land while it runs (pipeline-fill
// latency and the workgroup hand-over gap, ~3 us of a 26-32 us tile at K = 768, leave the critical
// path).
template <int EPI>
__global__ __launch_bounds__(512, 2) void gemm_nt_persist_kernel(GemmNtArgs p) {
  using C = Cfg<64, 2, 2, 4, 8>;
  constexpr int BM = C::BM, BN = C::BN, BK = 64, WN = 4;
  constexpr int A_BYTES = C::A_BYTES, STAGE_BYTES = C::STAGE_BYTES, ROW_B = C::ROW_B, CHUNKS = C::CHUNKS;
  constexpr int RPI = C::ROWS_PER_INSTR, NA = C::NA, NB = C::NB, KS = C::KS;
  constexpr int NWAVES = C::NWAVES, MT = C::MT, WROWS = C::MT * 16;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ntiles = p.tiles_m * p.tiles_n;
  const int grid = gridDim.x;

  // With 128-byte rows (8 chunks) the swizzle (row >> 1) & 7 of staging instruction i and of
  // fragment row-block i does not depend on i (their row offsets are multiples of 16), so one
  // per-lane offset serves every instruction; the per-instruction part is a uniform constant.
  static_assert(CHUNKS == 8 && RPI == 8, "persistent kernel is written for BK = 64");
  const int srow = wave * RPI + lane / CHUNKS;                       // staging row of instruction 0
  const int kc = ((lane % CHUNKS) ^ swz_row(srow, CHUNKS)) * 8;      // first k of this lane's chunk
  const uint32_t offA0 = (uint32_t)(srow * p.lda * 2 + kc * 2);
  const uint32_t offB0 = (uint32_t)(srow * p.ldb * 2 + kc * 2);
  const uint32_t stepA = (uint32_t)(NWAVES * RPI) * (uint32_t)p.lda * 2u;  // 64 rows further
  const uint32_t stepB = (uint32_t)(NWAVES * RPI) * (uint32_t)p.ldb * 2u;
  const int wm = wave / WN, wn = wave % WN;
  const int c16 = lane & 15, q4 = lane >> 4;
  uint32_t fx0[KS], fw0[KS];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    const int rx = wm * WROWS + c16, rw = wn * 64 + c16;
    fx0[ks] = rx * ROW_B + (((ks * 4 + q4) ^ swz_row(rx, CHUNKS)) << 4);
    fw0[ks] = rw * ROW_B + (((ks * 4 + q4) ^ swz_row(rw, CHUNKS)) << 4);
  }
  float* st = reinterpret_cast<float*>(lds + 2 * STAGE_BYTES) + wave * (16 * 64);
  const int nk = (p.K + BK - 1) / BK;

  // round r: workgroup w takes logical tile r * grid + remap(w) (bijective inside the round, the
  // workgroups of one XCD get neighbouring tiles), grouped GROUP_M row-tiles x all column tiles
  auto tile_origin = [&](int round, int& m0, int& n0) -> bool {
    const int base = round * grid;
    const int left = ntiles - base;
    if (left <= 0) return false;
    const int nthis = min(left, grid);
    if ((int)blockIdx.x >= nthis) return false;
    const int t = base + xcd_remap(blockIdx.x, nthis);
    const int gsz = GROUP_M * p.tiles_n;
    const int grp = t / gsz;
    const int first_m = grp * GROUP_M;
    const int gm = min(GROUP_M, p.tiles_m - first_m);
    const int in_g = t - grp * gsz;
    m0 = (first_m + in_g % gm) * BM;
    n0 = (in_g / gm) * BN;
    return true;
  };
  auto stage = [&](const __amdgpu_buffer_rsrc_t& rsA, const __amdgpu_buffer_rsrc_t& rsB, int buf, int k0) {
    char* la = lds + buf * STAGE_BYTES;
    char* lb = la + A_BYTES;
    const bool in_k = (k0 + kc < p.K);
    // opaque copies: keeps the compiler from hoisting the eight per-instruction offsets out of the
    // K loop as loop invariants (they would be spilled; one v_add each is cheaper)
    uint32_t oa = offA0, ob = offB0;
    asm volatile("" : "+v"(oa), "+v"(ob));
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      const uint32_t va = in_k ? oa + (uint32_t)i * stepA + (uint32_t)k0 * 2 : MVPTR_OOB;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, LDS_PTR(la + (i * NWAVES + wave) * 1024), 16, va, 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      const uint32_t vb = in_k ? ob + (uint32_t)i * stepB + (uint32_t)k0 * 2 : MVPTR_OOB;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, LDS_PTR(lb + (i * NWAVES + wave) * 1024), 16, vb, 0, 0, 0);
    }
  };
  auto rsrc_a = [&](int m0) {
    const int rows_a = min(BM, p.M - m0);
    return make_rsrc(p.A + (int64_t)m0 * p.lda, (uint32_t)(((int64_t)(rows_a - 1) * p.lda + p.K) * 2));
  };
  auto rsrc_b = [&](int n0) {
    const int rows_b = min(BN, p.N - n0);
    return make_rsrc(p.B + (int64_t)n0 * p.ldb, (uint32_t)(((int64_t)(rows_b - 1) * p.ldb + p.K) * 2));
  };

  int m0 = 0, n0 = 0;
  if (!tile_origin(0, m0, n0)) return;
  __amdgpu_buffer_rsrc_t rsA = rsrc_a(m0), rsB = rsrc_b(n0);
  bool prefetched = false;
  for (int round = 0;; ++round) {
    f32x4 acc[4][MT];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < MT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (!prefetched) stage(rsA, rsB, 0, 0);
    for (int kt = 0; kt < nk; ++kt) {
      // double buffer: K-step kt has landed when nothing is outstanding (the epilogue's stores of
      // the previous tile included); the barrier also frees the other buffer for K-step kt + 1
      asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
      const int buf = kt & 1;
      const char* la = lds + buf * STAGE_BYTES;
      const char* lb = la + A_BYTES;
      bf16x8 xf[MT], wf[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) wf[i] = *reinterpret_cast<const bf16x8*>(lb + fw0[0] + i * 16 * ROW_B);
#pragma unroll
      for (int i = 0; i < MT; ++i) xf[i] = *reinterpret_cast<const bf16x8*>(la + fx0[0] + i * 16 * ROW_B);
      if (kt + 1 < nk && !(prefetched && kt == 0)) stage(rsA, rsB, buf ^ 1, (kt + 1) * BK);
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        if (ks > 0) {
#pragma unroll
          for (int i = 0; i < 4; ++i) wf[i] = *reinterpret_cast<const bf16x8*>(lb + fw0[ks] + i * 16 * ROW_B);
#pragma unroll
          for (int i = 0; i < MT; ++i) xf[i] = *reinterpret_cast<const bf16x8*>(la + fx0[ks] + i * 16 * ROW_B);
        }
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
          for (int mt = 0; mt < MT; ++mt)
            acc[nt][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[nt], xf[mt], acc[nt][mt], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
      }
    }
    __syncthreads();  // every wave is done with the operand ring
    int m1 = 0, n1 = 0;
    const bool more = tile_origin(round + 1, m1, n1);
    const bool pf = more && !(p.exp_flags & 1);
    __amdgpu_buffer_rsrc_t rsA1 = rsA, rsB1 = rsB;
    if (more) {
      rsA1 = rsrc_a(m1);
      rsB1 = rsrc_b(n1);
    }
    if (pf) {
      stage(rsA1, rsB1, 0, 0);
      if (nk > 1) stage(rsA1, rsB1, 1, BK);
    }
    // the epilogue's lane-derived constants are recomputed per tile (opaque lane copy) instead of
    // living in registers across the MFMA loop
    int lane_e = lane;
    asm volatile("" : "+v"(lane_e));
    nt_epilogue<EPI, MT, 16>(p, acc, st, m0, n0, wm, wn, lane_e);
    if (!more) break;
    m0 = m1;
    n0 = n1;
    rsA = rsA1;
    rsB = rsB1;
    prefetched = pf;
  }
}

// ---------------------------------------------------------------------------------------------
// "Q" configuration: 256x256 tile per 256-thread workgroup, FOUR waves as 2(M) x 2(N), each
// 128x128 = 4x4 v_mfma_f32_32x32x16_bf16 (256 accumulator registers; one wave per SIMD owns the
// whole 512-register file).  Per MFMA: half a ds_read_b128 (8 fragment reads per 16 MFMAs) and the
// fewest L2->LDS bytes per FLOP of any tile here.
//  * 64 k per stage (whole 128-byte lines per row: half-line fetches of a 32-k stage measured 10 %
//    slower), 64 KiB per stage, double buffer; a row's eight 16-byte chunks XOR-swizzled with
//    (row >> 1) & 7: the 32x32x16 operand reads (a lane reads chunk 2s+h of row r) are conflict free.
//  * the LDS-DMA loads are hand-issued (lds_dma16): hipcc drains the builtin form with a vmcnt(0)
//    in front of the next ds_read, which serialises a ring; here every wait is the counted one in
//    front of the barrier.
//  * rotated loop (see gemm_tn_q_kernel): the barrier sits in front of the last of a stage's four
//    16-k sub-steps; the LDS-DMA issue of the stage after next and the fragment reads of the next
//    stage run under that sub-step's 16 MFMAs; quarters pinned with sched_barrier(0).
//  * the weight tile is the MFMA A operand: a lane holds 4 consecutive output columns per register
//    group; v_permlane32_swap pairs two groups so that each lane owns 8 consecutive columns of one
//    row and the epilogue finishes straight from registers with 16-byte loads / stores (no LDS
//    restaging; a row's 128 columns are written by 4 consecutive store instructions).
template <int EPI>
__device__ __forceinline__ void ntq_finish8(const GemmNtArgs& p, int m, int n, float (&v)[8]) {
  // v = 8 consecutive columns n..n+7 of row m of the f32 product; N % 8 == 0 (launch condition)
  if (m >= p.M || n >= p.N) return;
  if ((p.exp_flags & 4) && v[0] != 12345.678f) return;  // ablation (MVPTR_NT_EXP bit 2): no loads / math / stores
  if (EPI != MVPTR_EPI_GELU_BWD && EPI != MVPTR_EPI_ADD && p.bias != nullptr) {
    const f32x4 b0 = *reinterpret_cast<const f32x4*>(p.bias + n);
    const f32x4 b1 = *reinterpret_cast<const f32x4*>(p.bias + n + 4);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      v[e] += b0[e];
      v[4 + e] += b1[e];
    }
  }
  constexpr bool kNeedsAux = (EPI == MVPTR_EPI_BIAS_RESID || EPI == MVPTR_EPI_GELU_BWD || EPI == MVPTR_EPI_ADD);
  float a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (kNeedsAux && p.aux != nullptr) {
    const bf16x8 x = *reinterpret_cast<const bf16x8*>(p.aux + (int64_t)m * p.ld_aux + n);
#pragma unroll
    for (int e = 0; e < 8; ++e) a[e] = bf2f(x[e]);
  }
  auto store_bf8 = [&](void* base, const float (&o)[8]) {
    bf16x8 t;
#pragma unroll
    for (int e = 0; e < 8; ++e) t[e] = f2bf(o[e]);
    *reinterpret_cast<bf16x8*>((__bf16*)base + (int64_t)m * p.ldc + n) = t;
  };
  if (EPI == MVPTR_EPI_BIAS) {
    store_bf8(p.out0, v);
  } else if (EPI == MVPTR_EPI_BIAS_GELU) {
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
    store_bf8(p.out0, dg);
    store_bf8(p.out1, g);
  } else if (EPI == MVPTR_EPI_BIAS_RESID) {
#pragma unroll
    for (int e = 0; e < 8; e += 2)  // N even: (m * N + n + e) even, lanes own whole hash pairs
      drop_apply2(p.drop, (uint64_t)m * (uint64_t)p.N + (uint64_t)(n + e), v[e], v[e + 1]);
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] += a[e];
    store_bf8(p.out0, v);
  } else if (EPI == MVPTR_EPI_GELU_BWD) {
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] *= a[e];
    store_bf8(p.out0, v);
  } else if (EPI == MVPTR_EPI_ADD) {
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] += a[e];
    store_bf8(p.out0, v);
  } else if (EPI == MVPTR_EPI_F32) {
    float* op = (float*)p.out0 + (int64_t)m * p.ldc + n;
    *reinterpret_cast<f32x4*>(op) = f32x4{v[0], v[1], v[2], v[3]};
    *reinterpret_cast<f32x4*>(op + 4) = f32x4{v[4], v[5], v[6], v[7]};
  } else if (EPI == MVPTR_EPI_BIAS_TANH) {
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = tanhf(v[e]);
    store_bf8(p.out0, v);
  }
}

template <int EPI>
__global__ __launch_bounds__(256, 1) void gemm_ntq_kernel(GemmNtArgs p) {
  constexpr int BM = 256, BN = 256, BK = 64;
  constexpr int OP_B = 256 * BK * 2;      // one operand tile of a stage (32 KiB)
  constexpr int STAGE_B = 2 * OP_B;       // activations, then weights
  constexpr int LPS = 16;                 // LDS-DMA instructions per wave and stage
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nwg = p.tiles_m * p.tiles_n;
  const int t = xcd_remap(blockIdx.x, nwg);
  const int gsz = GROUP_M * p.tiles_n;
  const int grp = t / gsz;
  const int first_m = grp * GROUP_M;
  const int gm = min(GROUP_M, p.tiles_m - first_m);
  const int in_g = t - grp * gsz;
  const int tm = first_m + in_g % gm;
  const int tn = in_g / gm;
  const int m0 = tm * BM, n0 = tn * BN;
  const int rows_a = min(BM, p.M - m0);
  const int rows_b = min(BN, p.N - n0);
  const u32x4 rsA = make_rsrc_words(p.A + (int64_t)m0 * p.lda, (uint32_t)(((int64_t)(rows_a - 1) * p.lda + p.K) * 2));
  const u32x4 rsB = make_rsrc_words(p.B + (int64_t)n0 * p.ldb, (uint32_t)(((int64_t)(rows_b - 1) * p.ldb + p.K) * 2));
  const uint32_t lds0 = lds_addr(lds);

  // staging instruction i (0..7) of this wave fills KiB (i * 4 + wave) of an operand tile: rows
  // 8 * (i * 4 + wave) .. + 7, whole 128-byte lines; lane -> row + (lane >> 3), 16-byte slot
  // lane & 7 holds k-chunk slot ^ ((row >> 1) & 7) = slot ^ ((4 * (wave & 1) + (lane >> 4)) & 7)
  const int srow = wave * 8 + (lane >> 3);
  const int kc = ((lane & 7) ^ ((4 * (wave & 1) + (lane >> 4)) & 7)) * 8;
  const uint32_t offA0 = (uint32_t)(srow * p.lda * 2 + kc * 2);
  const uint32_t offB0 = (uint32_t)(srow * p.ldb * 2 + kc * 2);
  const uint32_t rstepA = (uint32_t)(32 * p.lda * 2), rstepB = (uint32_t)(32 * p.ldb * 2);
  auto stage_piece = [&](int buf, int st, int i) {
    const uint32_t la = lds0 + (uint32_t)(buf * STAGE_B + wave * 1024);
    const bool in_k = (st * BK + kc < p.K);
    if (i < 8) {
      const uint32_t va = in_k ? offA0 + (uint32_t)i * rstepA + (uint32_t)(st * BK * 2) : MVPTR_OOB;
      lds_dma16(rsA, va, la + i * 4096);
    } else {
      const uint32_t vb = in_k ? offB0 + (uint32_t)(i - 8) * rstepB + (uint32_t)(st * BK * 2) : MVPTR_OOB;
      lds_dma16(rsB, vb, la + OP_B + (i - 8) * 4096);
    }
  };

  const int wm = wave >> 1, wn = wave & 1;
  const int r31 = lane & 31, h = lane >> 5;
  // fragment offsets of the four 16-k sub-steps; row block b adds b * 4096 (an immediate)
  uint32_t fx[4], fw[4];
#pragma unroll
  for (int s4 = 0; s4 < 4; ++s4) {
    const uint32_t o = (uint32_t)(r31 * 128 + (((2 * s4 + h) ^ ((r31 >> 1) & 7)) << 4));
    fx[s4] = (uint32_t)(wm * 128 * 128) + o;
    fw[s4] = (uint32_t)(OP_B + wn * 128 * 128) + o;
  }

  f32x16 acc[4][4];  // [nb][mb]
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  bf16x8 wf0[4], xf0[4], wf1[4], xf1[4];
  // quarter nb multiplies W fragment nb with all four X fragments.  The eight fragment reads of a
  // sub-step are spread X0 X1 X2 | X3 W0 W1 | W2 W3 | - over the quarters of the previous sub-step:
  // the last quarter issues none, so they have all landed when the next sub-step's first MFMA waits
  auto read_frag = [&](const char* base, int s4, int f, bf16x8(&wf)[4], bf16x8(&xf)[4]) {
    if (f < 4) xf[f] = *reinterpret_cast<const bf16x8*>(base + fx[s4] + f * 4096);
    else wf[f - 4] = *reinterpret_cast<const bf16x8*>(base + fw[s4] + (f - 4) * 4096);
  };
  auto read_quarter = [&](const char* base, int s4, int j, bf16x8(&wf)[4], bf16x8(&xf)[4]) {
    constexpr int first[5] = {0, 3, 6, 8, 8};
#pragma unroll
    for (int f = 0; f < 8; ++f)
      if (f >= first[j] && f < first[j + 1]) read_frag(base, s4, f, wf, xf);
  };
  auto mma_row = [&](int nb, const bf16x8(&wf)[4], const bf16x8(&xf)[4]) {
#pragma unroll
    for (int mb = 0; mb < 4; ++mb)
      acc[nb][mb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[nb], xf[mb], acc[nb][mb], 0, 0, 0);
  };

  const int nsteps = (p.K + BK - 1) / BK;
#pragma unroll
  for (int j = 0; j < 16; ++j) stage_piece(0, 0, j);
  if (nsteps > 1) {
#pragma unroll
    for (int j = 0; j < 16; ++j) stage_piece(1, 1, j);
    asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(LPS) : "memory");
  } else {
    asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) read_quarter(lds, 0, j, wf0, xf0);
  // One 64-k stage = four 16-k sub-steps of 16 MFMAs; sub-step s multiplies the fragments read
  // during sub-step s-1.  The barrier sits in front of the LAST sub-step: by then this wave has read
  // all of stage st (lgkmcnt(0)) and the next stage has landed (it is the only one in flight), so the
  // last sub-step's MFMAs cover the LDS-DMA issue of stage st+2 into the buffer just freed and the
  // reads of the next stage's first fragments.
  // MORE / REFILL are compile-time tags (run-time branches around the MFMA quarters make hipcc spill).
  auto step = [&](int st, auto more_tag, auto refill_tag) {
    constexpr bool MORE = decltype(more_tag)::value, REFILL = decltype(refill_tag)::value;
    const char* cur = lds + (st & 1) * STAGE_B;
    const char* nxt = lds + ((st + 1) & 1) * STAGE_B;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      read_quarter(cur, 1, j, wf1, xf1);
      mma_row(j, wf0, xf0);
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      read_quarter(cur, 2, j, wf0, xf0);
      mma_row(j, wf1, xf1);
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      read_quarter(cur, 3, j, wf1, xf1);
      mma_row(j, wf0, xf0);
      __builtin_amdgcn_sched_barrier(0);
    }
    if constexpr (MORE) {
      __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0): this wave's reads of stage st are done
      asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if constexpr (REFILL) {
#pragma unroll
        for (int i = 0; i < 4; ++i) stage_piece(st & 1, st + 2, 4 * j + i);
      }
      if constexpr (MORE) read_quarter(nxt, 0, j, wf0, xf0);
      mma_row(j, wf1, xf1);
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  {
    int st = 0;
    for (; st + 2 < nsteps; ++st) step(st, std::true_type{}, std::true_type{});
    if (st + 1 < nsteps) step(st++, std::true_type{}, std::false_type{});
    step(st, std::false_type{}, std::false_type{});
  }

  if (p.exp_flags & 2) return;  // ablation (MVPTR_NT_EXP bit 1): no epilogue at all
  // epilogue straight from the accumulators.  Block (nb, mb): lane (r31, h) holds row m = ..+r31,
  // columns 8g + 4h + (0..3) in registers 4g..4g+3; swapping group 2q+1 of the low half-wave with
  // group 2q of the high half-wave leaves columns 16q + 8h + (0..7) in each lane.
  int lane_e = lane;
  asm volatile("" : "+v"(lane_e));
  const int er = lane_e & 31, eh = lane_e >> 5;
#pragma unroll
  for (int mb = 0; mb < 4; ++mb) {
    const int m = m0 + wm * 128 + mb * 32 + er;
#pragma unroll
    for (int nb = 0; nb < 4; ++nb)
#pragma unroll
      for (int q2 = 0; q2 < 2; ++q2) {
        float v[8];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float lo = acc[nb][mb][8 * q2 + e], hi = acc[nb][mb][8 * q2 + 4 + e];
          // lanes 32-63 of `lo` <-> lanes 0-31 of `hi` (two wait states after a VALU write of either)
          asm("v_nop\n\tv_nop\n\tv_permlane32_swap_b32 %0, %1" : "+v"(lo), "+v"(hi));
          v[e] = lo;
          v[4 + e] = hi;
        }
        const int n = n0 + wn * 128 + nb * 32 + 16 * q2 + 8 * eh;
        ntq_finish8<EPI>(p, m, n, v);
      }
  }
}

// ---------------------------------------------------------------------------------------------
// "QP": the Q tile as a PERSISTENT kernel with a DEFERRED epilogue.  With K = 768 a 256x256 tile is
// ~20 us of MFMA work and 128-256 KiB of output; when every CU finishes its tile at the same time
// the outputs leave as one burst at the chip's write rate (~3 TB/s: 100 us of a 300-us launch,
// measured with the stores ablated) while no MFMA runs.  Here a workgroup walks over its tiles with
// one continuous LDS-DMA / MFMA pipeline; at the end of a tile the accumulators (+ bias) are packed
// to bf16 into 128 "pending" registers, and the epilogue proper (permlane swaps to 8-column runs,
// GELU, the 16-byte stores) is dripped through the first eight 64-k stages of the NEXT tile, one
// 32-row x 128-byte group per stage, under that tile's MFMAs.  Output writes are thereby spread
// evenly over the launch and overlap the matrix pipe chip-wide.
//  * stage loads run two stages ahead across tile boundaries (issue-side tile state in SGPRs); after
//    the last tile the descriptors have zero records, so the tail needs no branches.
//  * the first MFMA of every accumulator block of a tile takes C = 0: no accumulator clearing.
//  * the bias row of a tile is staged in LDS (one LDS-DMA instruction of wave 0 in the tile's first
//    stage) and added in f32 before the bf16 rounding: EPI_BIAS results are bit-identical to the
//    other configurations.
//  * needs K >= 512 (eight stages to drip into) and the Q conditions; epilogues without an aux operand.
struct NtqpOut {
  __amdgpu_buffer_rsrc_t out0, out1;
};
// Every call issues the same number of store instructions whatever the lane predicate (rows /
// columns outside the problem, or no pending tile, take the always-out-of-range offset): the
// barrier waits of the main loop count them.
template <int EPI>
__device__ __forceinline__ void ntqp_finish8(const GemmNtArgs& p, const NtqpOut& o, int m, int n, const u32x4& pk) {
  // pk = 8 consecutive columns n..n+7 of row m, bf16(acc + bias)
  const uint32_t off = (m < p.M && n < p.N) ? (uint32_t)(((int64_t)m * p.ldc + n) * 2) : MVPTR_OOB;
  if (EPI == MVPTR_EPI_BIAS) {
    __builtin_amdgcn_raw_buffer_store_b128(pk, o.out0, off, 0, 0);
  } else if (EPI == MVPTR_EPI_BIAS_GELU) {
    u32x4 og, od;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float u0 = __builtin_bit_cast(float, pk[e] << 16);
      const float u1 = __builtin_bit_cast(float, pk[e] & 0xffff0000u);
      f32x2 a2, d2;
      gelu_pair(f32x2{u0, u1}, a2, d2);
      const bf16x2 ga = {f2bf(a2.x), f2bf(a2.y)};
      const bf16x2 gd = {f2bf(d2.x), f2bf(d2.y)};
      og[e] = __builtin_bit_cast(uint32_t, ga);
      od[e] = __builtin_bit_cast(uint32_t, gd);
    }
    __builtin_amdgcn_raw_buffer_store_b128(od, o.out0, off, 0, 0);
    __builtin_amdgcn_raw_buffer_store_b128(og, o.out1, off, 0, 0);
  }
}
template <int EPI>
constexpr int ntqp_stores_per_unit() { return EPI == MVPTR_EPI_BIAS_GELU ? 2 : 1; }

template <int EPI>
__global__ __launch_bounds__(256, 1) void gemm_ntqp_kernel(GemmNtArgs p) {
  constexpr int BM = 256, BN = 256, BK = 64;
  constexpr int OP_B = 256 * BK * 2;      // one operand tile of a stage (32 KiB)
  constexpr int STAGE_B = 2 * OP_B;       // activations, then weights
  constexpr int BIAS_OFF = 2 * STAGE_B;   // two 1-KiB bias rows (tile parity)
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ntiles = p.tiles_m * p.tiles_n;
  const int grid = gridDim.x;
  const int nk = (p.K + BK - 1) / BK;
  const uint32_t lds0 = lds_addr(lds);

  auto tile_origin = [&](int round, int& m0, int& n0) -> bool {
    const int base = round * grid;
    const int left = ntiles - base;
    if (left <= 0) return false;
    const int nthis = min(left, grid);
    if ((int)blockIdx.x >= nthis) return false;
    const int t = base + xcd_remap(blockIdx.x, nthis);
    const int gsz = GROUP_M * p.tiles_n;
    const int grp = t / gsz;
    const int first_m = grp * GROUP_M;
    const int gm = min(GROUP_M, p.tiles_m - first_m);
    const int in_g = t - grp * gsz;
    m0 = (first_m + in_g % gm) * BM;
    n0 = (in_g / gm) * BN;
    return true;
  };

  // ---- issue side: the tile whose stages are being loaded (two stages ahead of the MFMAs)
  int is_round = 0, is_k = 0;
  u32x4 rsA, rsB;
  auto issue_tile = [&](int round) {
    int m0 = 0, n0 = 0;
    if (tile_origin(round, m0, n0)) {
      const int rows_a = min(BM, p.M - m0), rows_b = min(BN, p.N - n0);
      rsA = make_rsrc_words(p.A + (int64_t)m0 * p.lda, (uint32_t)(((int64_t)(rows_a - 1) * p.lda + p.K) * 2));
      rsB = make_rsrc_words(p.B + (int64_t)n0 * p.ldb, (uint32_t)(((int64_t)(rows_b - 1) * p.ldb + p.K) * 2));
    } else {
      rsA = make_rsrc_words(p.A, 0u);  // zero records: every load returns zeros without touching memory
      rsB = make_rsrc_words(p.B, 0u);
    }
  };
  issue_tile(0);
  const int srow = wave * 8 + (lane >> 3);
  const int kc = ((lane & 7) ^ ((4 * (wave & 1) + (lane >> 4)) & 7)) * 8;
  const uint32_t offA0 = (uint32_t)(srow * p.lda * 2 + kc * 2);
  const uint32_t offB0 = (uint32_t)(srow * p.ldb * 2 + kc * 2);
  const uint32_t rstepA = (uint32_t)(32 * p.lda * 2), rstepB = (uint32_t)(32 * p.ldb * 2);
  auto stage_piece = [&](int buf, int i) {
    const uint32_t la = lds0 + (uint32_t)(buf * STAGE_B + wave * 1024);
    const bool in_k = (is_k * BK + kc < p.K);
    if (i < 8) {
      const uint32_t va = in_k ? offA0 + (uint32_t)i * rstepA + (uint32_t)(is_k * BK * 2) : MVPTR_OOB;
      lds_dma16(rsA, va, la + i * 4096);
    } else {
      const uint32_t vb = in_k ? offB0 + (uint32_t)(i - 8) * rstepB + (uint32_t)(is_k * BK * 2) : MVPTR_OOB;
      lds_dma16(rsB, vb, la + OP_B + (i - 8) * 4096);
    }
  };
  auto issue_advance = [&]() {
    if (++is_k == nk) {
      is_k = 0;
      issue_tile(++is_round);
    }
  };

  const int wm = wave >> 1, wn = wave & 1;
  const int r31 = lane & 31, h = lane >> 5;
  uint32_t fx[4], fw[4];
#pragma unroll
  for (int s4 = 0; s4 < 4; ++s4) {
    const uint32_t o = (uint32_t)(r31 * 128 + (((2 * s4 + h) ^ ((r31 >> 1) & 7)) << 4));
    fx[s4] = (uint32_t)(wm * 128 * 128) + o;
    fw[s4] = (uint32_t)(OP_B + wn * 128 * 128) + o;
  }

  f32x16 acc[4][4];    // [nb][mb]
  uint32_t pend[4][4][8];  // [nb][mb][register pair]: bf16(acc + bias) of the previous tile
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int e = 0; e < 8; ++e) pend[i][j][e] = 0u;
  int pm0 = 0x40000000, pn0 = 0;  // origin of the pending tile (none yet: no row passes m < M)

  bf16x8 wf0[4], xf0[4], wf1[4], xf1[4];
  auto read_frag = [&](const char* base, int s4, int f, bf16x8(&wf)[4], bf16x8(&xf)[4]) {
    if (f < 4) xf[f] = *reinterpret_cast<const bf16x8*>(base + fx[s4] + f * 4096);
    else wf[f - 4] = *reinterpret_cast<const bf16x8*>(base + fw[s4] + (f - 4) * 4096);
  };
  auto read_quarter = [&](const char* base, int s4, int j, bf16x8(&wf)[4], bf16x8(&xf)[4]) {
    constexpr int first[5] = {0, 3, 6, 8, 8};
#pragma unroll
    for (int f = 0; f < 8; ++f)
      if (f >= first[j] && f < first[j + 1]) read_frag(base, s4, f, wf, xf);
  };
  auto mma_row = [&](int nb, const bf16x8(&wf)[4], const bf16x8(&xf)[4], auto first_tag) {
#pragma unroll
    for (int mb = 0; mb < 4; ++mb) {
      if constexpr (decltype(first_tag)::value) {
        const f32x16 z = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        acc[nb][mb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[nb], xf[mb], z, 0, 0, 0);
      } else {
        acc[nb][mb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[nb], xf[mb], acc[nb][mb], 0, 0, 0);
      }
    }
  };
  const u32x4 rsBias = make_rsrc_words(p.bias, p.bias != nullptr ? (uint32_t)p.N * 4u : 0u);
  const uint32_t out_bytes = (uint32_t)(((int64_t)(p.M - 1) * p.ldc + p.N) * 2);  // < 4 GiB (launch condition)
  NtqpOut outs;
  outs.out0 = make_rsrc_uniform(p.out0, out_bytes);
  outs.out1 = make_rsrc_uniform(p.out1 != nullptr ? p.out1 : p.out0, p.out1 != nullptr ? out_bytes : 0u);
  // one unit of the deferred epilogue: block (nb, mb), register groups 2*q2 / 2*q2+1 -> the lane's
  // 8-column run at columns 32 nb + 16 q2 + 8 h of row 32 mb + r31
  auto drip_unit = [&](int nb, int mb, int q2) {
    u32x4 pk;
    uint32_t lo0 = pend[nb][mb][4 * q2], lo1 = pend[nb][mb][4 * q2 + 1];
    uint32_t hi0 = pend[nb][mb][4 * q2 + 2], hi1 = pend[nb][mb][4 * q2 + 3];
    asm("v_nop\n\tv_nop\n\tv_permlane32_swap_b32 %0, %1" : "+v"(lo0), "+v"(hi0));
    asm("v_nop\n\tv_nop\n\tv_permlane32_swap_b32 %0, %1" : "+v"(lo1), "+v"(hi1));
    pk[0] = lo0;
    pk[1] = lo1;
    pk[2] = hi0;
    pk[3] = hi1;
    int lane_e = lane;
    asm volatile("" : "+v"(lane_e));
    const int m = pm0 + wm * 128 + mb * 32 + (lane_e & 31);
    const int n = pn0 + wn * 128 + nb * 32 + 16 * q2 + 8 * (lane_e >> 5);
    ntqp_finish8<EPI>(p, outs, m, n, pk);
  };
  // group g (0..7) = rows 32 (g >> 1).., columns 64 (g & 1)..: four units = whole 128-byte lines
  auto drip_group_unit = [&](int g, int u) { drip_unit(2 * (g & 1) + (u >> 1), g >> 1, u & 1); };


  int round = 0, m0 = 0, n0 = 0;
  tile_origin(0, m0, n0);  // grid <= ntiles: always valid
  // prologue: two stages in flight, the first one landed
#pragma unroll
  for (int j = 0; j < 16; ++j) stage_piece(0, j);
  issue_advance();
#pragma unroll
  for (int j = 0; j < 16; ++j) stage_piece(1, j);
  issue_advance();
  asm volatile("s_waitcnt vmcnt(16)\n\ts_barrier" ::: "memory");
#pragma unroll
  for (int j = 0; j < 4; ++j) read_quarter(lds, 0, j, wf0, xf0);
  int bufsel = 0;

  // G: drip group handled in this stage (-1: none); FIRST: the tile's first stage (C = 0 MFMAs and
  // the bias-row load)
  auto stage_body = [&](auto g_tag, auto first_tag, auto second_tag) {
    constexpr int G = decltype(g_tag)::value;
    constexpr bool FIRST = decltype(first_tag)::value;
    (void)second_tag;
    const char* cur = lds + bufsel * STAGE_B;
    const char* nxt = lds + (bufsel ^ 1) * STAGE_B;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      read_quarter(cur, 1, j, wf1, xf1);
      mma_row(j, wf0, xf0, first_tag);
      if constexpr (G >= 0) {
        if (j == 0) drip_group_unit(G, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      read_quarter(cur, 2, j, wf0, xf0);
      mma_row(j, wf1, xf1, std::false_type{});
      if constexpr (G >= 0) {
        if (j == 0) drip_group_unit(G, 1);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      read_quarter(cur, 3, j, wf1, xf1);
      mma_row(j, wf0, xf0, std::false_type{});
      if constexpr (G >= 0) {
        if (j == 0) drip_group_unit(G, 2);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0): this wave's reads of the stage are done
    // the next stage's loads are older than this stage's deferred stores, which may stay in flight
    asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(G >= 0 ? 3 * ntqp_stores_per_unit<EPI>() : 0) : "memory");
#pragma unroll
    for (int j = 0; j < 4; ++j) {
#pragma unroll
      for (int i = 0; i < 4; ++i) stage_piece(bufsel, 4 * j + i);
      read_quarter(nxt, 0, j, wf0, xf0);
      mma_row(j, wf1, xf1, std::false_type{});
      if constexpr (G >= 0) {
        // the group's last unit goes out behind the stage's LDS-DMA issue: its stores are OLDER than
        // nothing the next barrier waits for except themselves (counted there as in-flight-allowed)
        if (j == 3) drip_group_unit(G, 3);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    if constexpr (FIRST) {
      // the tile's bias row (256 floats) goes to LDS by one LDS-DMA instruction of wave 0; the next
      // barrier's vmcnt(0) covers it, the conversion at the end of the tile reads it
      if (wave == 0)
        lds_dma16(rsBias, (uint32_t)((n0 + lane * 4) * 4), lds0 + (uint32_t)(BIAS_OFF + (round & 1) * 1024));
    }
    issue_advance();
    bufsel ^= 1;
  };

  for (;;) {
    stage_body(std::integral_constant<int, 0>{}, std::true_type{}, std::false_type{});
    stage_body(std::integral_constant<int, 1>{}, std::false_type{}, std::true_type{});
    stage_body(std::integral_constant<int, 2>{}, std::false_type{}, std::false_type{});
    stage_body(std::integral_constant<int, 3>{}, std::false_type{}, std::false_type{});
    stage_body(std::integral_constant<int, 4>{}, std::false_type{}, std::false_type{});
    stage_body(std::integral_constant<int, 5>{}, std::false_type{}, std::false_type{});
    stage_body(std::integral_constant<int, 6>{}, std::false_type{}, std::false_type{});
    stage_body(std::integral_constant<int, 7>{}, std::false_type{}, std::false_type{});
    for (int kk = 8; kk < nk; ++kk)
      stage_body(std::integral_constant<int, -1>{}, std::false_type{}, std::false_type{});
    // tile end: accumulators + bias -> bf16 pending registers
    // (speed ablations, wrong results: MVPTR_QP_EXP 1 no bias reads, 2 no accumulator reads, 3 no conversion)
#if !defined(MVPTR_QP_EXP) || MVPTR_QP_EXP != 3
    {
      const char* brow = lds + BIAS_OFF + (round & 1) * 1024 + (wn * 128 + 4 * h) * 4;
#pragma unroll
      for (int nb = 0; nb < 4; ++nb) {
        f32x4 b[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
#if defined(MVPTR_QP_EXP) && MVPTR_QP_EXP == 1
          b[g] = f32x4{0.f, 0.f, 0.f, 0.f};
          (void)brow;
#else
          b[g] = *reinterpret_cast<const f32x4*>(brow + (nb * 32 + 8 * g) * 4);
#endif
        }
#pragma unroll
        for (int mb = 0; mb < 4; ++mb) {
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const int r = 2 * e;
#if defined(MVPTR_QP_EXP) && MVPTR_QP_EXP == 2
            float a0 = __builtin_bit_cast(float, pend[nb][mb][e]), a1 = a0 * 0.5f;
#else
            const float a0 = acc[nb][mb][r], a1 = acc[nb][mb][r + 1];
#endif
            const bf16x2 t2 = {f2bf(a0 + b[r >> 2][r & 3]), f2bf(a1 + b[r >> 2][(r & 3) + 1])};
            pend[nb][mb][e] = __builtin_bit_cast(uint32_t, t2);
          }
          // one block at a time: hipcc otherwise reads all 256 accumulators first and spills the
          // loop's address registers to make room
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
#endif
    pm0 = m0;
    pn0 = n0;
    ++round;
    if (!tile_origin(round, m0, n0)) break;
  }
  // the last tile's epilogue
#pragma unroll
  for (int g = 0; g < 8; ++g)
#pragma unroll
    for (int u = 0; u < 4; ++u) drip_group_unit(g, u);
}

template <int EPI>
int launch_qp(GemmNtArgs a, hipStream_t s) {
  constexpr int LDS_BYTES = 2 * 2 * 256 * 64 * 2 + 2048;  // 128 KiB ring + two bias rows
  a.tiles_m = (a.M + 255) / 256;
  a.tiles_n = (a.N + 255) / 256;
  if ((int64_t)256 * a.lda * 2 >= (int64_t)0x7fffffff || (int64_t)256 * a.ldb * 2 >= (int64_t)0x7fffffff ||
      (int64_t)a.M * a.ldc * 2 >= (int64_t)0x7fffffff)
    MVPTR_FAIL(MVPTR_BAD_SHAPE, "gemm_nt: leading dimension too large");
  hipError_t e = hipFuncSetAttribute((const void*)gemm_ntqp_kernel<EPI>,
                                     hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
  if (e != hipSuccess) MVPTR_FAIL(MVPTR_HIP_ERROR, "gemm_nt: set LDS size: %s", hipGetErrorString(e));
  static int num_cu = 0;
  if (num_cu == 0) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess)
      MVPTR_FAIL(MVPTR_HIP_ERROR, "gemm_nt: no HIP device");
    num_cu = prop.multiProcessorCount;
  }
  const int ntiles = a.tiles_m * a.tiles_n;
  const int grid = ntiles < num_cu ? ntiles : num_cu;
  hipLaunchKernelGGL((gemm_ntqp_kernel<EPI>), dim3(grid), dim3(256), LDS_BYTES, s, a);
  MVPTR_CHECK_LAUNCH("gemm_nt");
  return MVPTR_OK;
}
template <int EPI>
constexpr bool qp_has_epilogue() { return EPI == MVPTR_EPI_BIAS || EPI == MVPTR_EPI_BIAS_GELU; }

template <int EPI>
int launch_q(GemmNtArgs a, hipStream_t s) {
  constexpr int LDS_BYTES = 2 * 2 * 256 * 64 * 2;  // 128 KiB
  a.tiles_m = (a.M + 255) / 256;
  a.tiles_n = (a.N + 255) / 256;
  if ((int64_t)256 * a.lda * 2 >= (int64_t)0x7fffffff || (int64_t)256 * a.ldb * 2 >= (int64_t)0x7fffffff)
    MVPTR_FAIL(MVPTR_BAD_SHAPE, "gemm_nt: leading dimension too large");
  hipError_t e = hipFuncSetAttribute((const void*)gemm_ntq_kernel<EPI>,
                                     hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
  if (e != hipSuccess) MVPTR_FAIL(MVPTR_HIP_ERROR, "gemm_nt: set LDS size: %s", hipGetErrorString(e));
  hipLaunchKernelGGL((gemm_ntq_kernel<EPI>), dim3(a.tiles_m * a.tiles_n), dim3(256), LDS_BYTES, s, a);
  MVPTR_CHECK_LAUNCH("gemm_nt");
  return MVPTR_OK;
}
// what the "Q" kernel's register epilogue needs: whole 8-column runs and 16-byte accesses
inline bool q_eligible(const GemmNtArgs& a) {
  if ((a.N & 7) || !a.vec_out_ok || a.vec_out != nullptr) return false;
  if (a.aux != nullptr && !a.vec_aux_ok) return false;
  if (a.bias != nullptr && !a.vec_bias_ok) return false;
  return true;
}

// ---------------------------------------------------------------------------------------------
// "pd": persistent kernel with a DEFERRED epilogue and TWO waves per SIMD.
//
// What bounds the K = 768 GEMMs: a CU can store ~24 GB/s, so the 128-256 KiB
// of a tile's outputs take 5-11 us during which the default kernel runs no MFMA; a one-wave-per-SIMD
// kernel ("qp") can overlap the stores but not the epilogue's VALU work with its own MFMAs.  Here the
// default kernel's loop (8 waves as 2 x 4, 16x16x32 MFMA, BK 64 double buffer, builtin LDS-DMA) runs
// persistently over 192 x 256 tiles: a wave owns 96 x 64 = 96 accumulator registers, which leaves room
// for the previous tile's outputs as 48 "pending" registers (bf16 of acc + bias).  At the end of a
// tile the next tile's first two K-steps are requested, the accumulators are packed into the pending
// registers (~200 VALU per wave) and the K-loop of the next tile starts; its first six K-steps each
// finish one 16-row block of the pending tile — unpack, restage through the wave's 4-KiB LDS area
// into row-chunk form, bias/GELU/dropout/residual math, 16-byte full-line stores — between their
// MFMAs, where the SIMD's other wave keeps the matrix pipe busy.
//  * vmcnt completes in issue order: a block's aux rows are requested BEFORE the K-step's LDS-DMA
//    issue (waiting for them then does not wait for the loads of the next K-step), its stores come
//    after it and may stay in flight across the next barrier (counted wait: every wave issues the
//    same number of store instructions per block, masked lanes take an out-of-range buffer offset).
//  * EPI_BIAS output is bit-identical to the default kernel (bias added in f32 before the rounding);
//    epilogues with an aux operand or GELU see one extra bf16 rounding of acc + bias.
//  * needs K >= 384 (six K-steps to drip into), N % 8 == 0 and 16-byte aligned operands.
template <int EPI>
__device__ __forceinline__ void nt_epilogue_aux(const GemmNtArgs& p, int mrow0, int ncol0, int lane, bf16x8 (&auxv)[2]) {
  constexpr bool kNeedsAux = (EPI == MVPTR_EPI_BIAS_RESID || EPI == MVPTR_EPI_GELU_BWD || EPI == MVPTR_EPI_ADD);
  if (!kNeedsAux) return;
  const int n = ncol0 + (lane & 7) * 8, rsub = lane >> 3;
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    const int mj = mrow0 + it * 8 + rsub;
    bf16x8 x;
#pragma unroll
    for (int e = 0; e < 8; ++e) x[e] = f2bf(0.f);
    if (p.aux != nullptr && mj < p.M && n < p.N) x = *reinterpret_cast<const bf16x8*>(p.aux + (int64_t)mj * p.ld_aux + n);
    auxv[it] = x;
  }
}
template <int EPI>
__device__ __forceinline__ void nt_epilogue_block(const GemmNtArgs& p, const NtqpOut& outs, const f32x4 (&blk)[4], float* st,
                                                  int mrow0, int ncol0, int lane, float (&cs)[8], const bf16x8 (&auxv)[2]) {
  // blk[nt] = 4 consecutive columns (16 nt + 4 (lane >> 4) ..) of row lane & 15 of a 16 x 64 block
  // whose bias has been added already; finished in row-chunk form (lane: row it*8 + lane>>3, 8 columns)
  const int c16 = lane & 15, q4 = lane >> 4;
  const int ch = lane & 7, rsub = lane >> 3;
#pragma unroll
  for (int nt = 0; nt < 4; ++nt)
    *reinterpret_cast<f32x4*>(st + c16 * 64 + (((nt * 4 + q4) ^ c16) << 2)) = blk[nt];
  const int n = ncol0 + ch * 8;
  constexpr bool kNeedsAux = (EPI == MVPTR_EPI_BIAS_RESID || EPI == MVPTR_EPI_GELU_BWD || EPI == MVPTR_EPI_ADD);
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    const int lrow = it * 8 + rsub;
    const int m = mrow0 + lrow;
    const f32x4 v0 = *reinterpret_cast<const f32x4*>(st + lrow * 64 + (((2 * ch) ^ lrow) << 2));
    const f32x4 v1 = *reinterpret_cast<const f32x4*>(st + lrow * 64 + (((2 * ch + 1) ^ lrow) << 2));
    // every lane runs the math and every wave issues the same number of store instructions (masked
    // lanes take the always-out-of-range offset): the K-loop's counted vmcnt relies on it
    const uint32_t off = (m < p.M && n < p.N) ? (uint32_t)(((int64_t)m * p.ldc + n) * 2) : MVPTR_OOB;
    float v[8], a[8];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      v[e] = v0[e];
      v[4 + e] = v1[e];
    }
    if (kNeedsAux) {
#pragma unroll
      for (int e = 0; e < 8; ++e) a[e] = bf2f(auxv[it][e]);
    }
    auto store_bf8 = [&](const __amdgpu_buffer_rsrc_t& rs, const float (&o)[8]) {
      bf16x8 t;
#pragma unroll
      for (int e = 0; e < 8; ++e) t[e] = f2bf(o[e]);
      // MVPTR_NT_EXP bit 9: sc1 (write-through, line not kept in the XCD's L2) instead of a plain store
      if (p.exp_flags & 512) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, t), rs, off, 0, 16);
      else __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, t), rs, off, 0, 0);
    };
    if (EPI == MVPTR_EPI_BIAS) {
      store_bf8(outs.out0, v);
    } else if (EPI == MVPTR_EPI_BIAS_GELU) {
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
      store_bf8(outs.out0, dg);
      store_bf8(outs.out1, g);
    } else if (EPI == MVPTR_EPI_BIAS_RESID) {
#pragma unroll
      for (int e = 0; e < 8; e += 2)
        drop_apply2(p.drop, (uint64_t)m * (uint64_t)p.N + (uint64_t)(n + e), v[e], v[e + 1]);
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] += a[e];
      store_bf8(outs.out0, v);
    } else if (EPI == MVPTR_EPI_GELU_BWD) {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        v[e] *= a[e];
        if (off != MVPTR_OOB) cs[e] += v[e];
      }
      store_bf8(outs.out0, v);
    } else if (EPI == MVPTR_EPI_ADD) {
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] += a[e];
      store_bf8(outs.out0, v);
    } else if (EPI == MVPTR_EPI_BIAS_TANH) {
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = tanhf(v[e]);
      store_bf8(outs.out0, v);
    }
  }
}

template <int EPI>
__global__ __launch_bounds__(512, 2) void gemm_nt_pd_kernel(GemmNtArgs p) {
  constexpr int BM = 192, BN = 256, BK = 64, WN = 4, MT = 6, WROWS = 96, NWAVES = 8;
  constexpr int A_BYTES = BM * BK * 2, B_BYTES = BN * BK * 2, STAGE_BYTES = A_BYTES + B_BYTES;
  constexpr int ROW_B = 128, NA = 3, NB = 4, KS = 2;
  constexpr int STORES_PER_BLOCK = (EPI == MVPTR_EPI_BIAS_GELU) ? 4 : 2;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ntiles = p.tiles_m * p.tiles_n;
  const int grid = gridDim.x;
  const int srow = wave * 8 + (lane >> 3);
  const int kc = ((lane & 7) ^ ((4 * wave + (lane >> 4)) & 7)) * 8;
  const uint32_t offA0 = (uint32_t)(srow * p.lda * 2 + kc * 2);
  const uint32_t offB0 = (uint32_t)(srow * p.ldb * 2 + kc * 2);
  const uint32_t stepA = 64u * (uint32_t)p.lda * 2u, stepB = 64u * (uint32_t)p.ldb * 2u;
  const int wm = wave / WN, wn = wave % WN;
  const int c16 = lane & 15, q4 = lane >> 4;
  uint32_t fx0[KS], fw0[KS];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    const int sw = (c16 >> 1) & 7;   // (row >> 1) & 7 of every fragment row this lane reads
    fx0[ks] = (uint32_t)((wm * WROWS + c16) * ROW_B + (((ks * 4 + q4) ^ sw) << 4));
    fw0[ks] = (uint32_t)((wn * 64 + c16) * ROW_B + (((ks * 4 + q4) ^ sw) << 4));
  }
  float* st = reinterpret_cast<float*>(lds + 2 * STAGE_BYTES) + wave * (16 * 64);
  const int nk = (p.K + BK - 1) / BK;

  auto tile_origin = [&](int round, int& m0, int& n0) -> bool {
    const int base = round * grid;
    const int left = ntiles - base;
    if (left <= 0) return false;
    const int nthis = min(left, grid);
    if ((int)blockIdx.x >= nthis) return false;
    const int t = base + xcd_remap(blockIdx.x, nthis);
    const int gsz = GROUP_M * p.tiles_n;
    const int grp = t / gsz;
    const int first_m = grp * GROUP_M;
    const int gm = min(GROUP_M, p.tiles_m - first_m);
    const int in_g = t - grp * gsz;
    m0 = (first_m + in_g % gm) * BM;
    n0 = (in_g / gm) * BN;
    return true;
  };
  auto stage = [&](const __amdgpu_buffer_rsrc_t& rsA, const __amdgpu_buffer_rsrc_t& rsB, int buf, int k0) {
    char* la = lds + buf * STAGE_BYTES;
    char* lb = la + A_BYTES;
    const bool in_k = (k0 + kc < p.K);
    uint32_t oa = offA0, ob = offB0;
    asm volatile("" : "+v"(oa), "+v"(ob));
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      const uint32_t va = in_k ? oa + (uint32_t)i * stepA + (uint32_t)k0 * 2 : MVPTR_OOB;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, LDS_PTR(la + (i * NWAVES + wave) * 1024), 16, va, 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      const uint32_t vb = in_k ? ob + (uint32_t)i * stepB + (uint32_t)k0 * 2 : MVPTR_OOB;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, LDS_PTR(lb + (i * NWAVES + wave) * 1024), 16, vb, 0, 0, 0);
    }
  };
  auto rsrc_a = [&](int m0) {
    const int rows_a = min(BM, p.M - m0);
    return make_rsrc(p.A + (int64_t)m0 * p.lda, (uint32_t)(((int64_t)(rows_a - 1) * p.lda + p.K) * 2));
  };
  auto rsrc_b = [&](int n0) {
    const int rows_b = min(BN, p.N - n0);
    return make_rsrc(p.B + (int64_t)n0 * p.ldb, (uint32_t)(((int64_t)(rows_b - 1) * p.ldb + p.K) * 2));
  };

  int m0 = 0, n0 = 0;
  if (!tile_origin(0, m0, n0)) return;
  __amdgpu_buffer_rsrc_t rsA = rsrc_a(m0), rsB = rsrc_b(n0);
  f32x4 acc[4][MT];
  uint32_t pend[MT][4][2];   // previous tile: bf16(acc + bias), [16-row block][16-column block][pair]
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) pend[i][j][0] = pend[i][j][1] = 0u;
  int pm0 = 0x40000000, pn0 = 0;  // origin of the pending tile (none yet: every row fails m < M)
  float cs[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  const uint32_t out_bytes = (uint32_t)(((int64_t)(p.M - 1) * p.ldc + p.N) * 2);  // < 4 GiB (launch condition)
  NtqpOut outs;
  outs.out0 = make_rsrc_uniform(p.out0, out_bytes);
  outs.out1 = make_rsrc_uniform(p.out1 != nullptr ? p.out1 : p.out0, p.out1 != nullptr ? out_bytes : 0u);

  // finish 16-row block `mt` of the pending tile: drip_load (aux rows) ahead of the K-step's LDS-DMA
  // issue, drip (math + stores) between its MFMAs
  bf16x8 auxv[2];
  auto drip_load = [&](int mt) {
    int lane_e = lane;
    asm volatile("" : "+v"(lane_e));
    nt_epilogue_aux<EPI>(p, pm0 + wm * WROWS + mt * 16, pn0 + wn * 64, lane_e, auxv);
  };
  auto drip = [&](int mt) {
    if (p.exp_flags & 128) {   // ablation (MVPTR_NT_EXP bit 7): no deferred epilogue work; keeps the store count
      const u32x4 z = {0u, 0u, 0u, 0u};
#pragma unroll
      for (int i = 0; i < STORES_PER_BLOCK; ++i) __builtin_amdgcn_raw_buffer_store_b128(z, outs.out0, MVPTR_OOB, 0, 0);
      return;
    }
    f32x4 blk[4];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
      blk[nt][0] = __builtin_bit_cast(float, pend[mt][nt][0] << 16);
      blk[nt][1] = __builtin_bit_cast(float, pend[mt][nt][0] & 0xffff0000u);
      blk[nt][2] = __builtin_bit_cast(float, pend[mt][nt][1] << 16);
      blk[nt][3] = __builtin_bit_cast(float, pend[mt][nt][1] & 0xffff0000u);
    }
    int lane_e = lane;
    asm volatile("" : "+v"(lane_e));   // epilogue constants re-derived here, not kept across the MFMA loop
    nt_epilogue_block<EPI>(p, outs, blk, st, pm0 + wm * WROWS + mt * 16, pn0 + wn * 64, lane_e, cs, auxv);
    if (EPI == MVPTR_EPI_GELU_BWD && mt == MT - 1 && p.vec_out != nullptr) {
      const int ch = lane_e & 7, rsub = lane_e >> 3;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        float sum = cs[e];
        sum += __shfl_xor(sum, 8);
        sum += __shfl_xor(sum, 16);
        sum += __shfl_xor(sum, 32);
        const int n = pn0 + wn * 64 + ch * 8 + e;
        if (rsub == 0 && n < p.N && pm0 < p.M) atomicAdd(p.vec_out + n, sum);
        cs[e] = 0.f;
      }
    }
  };

  bool prefetched = false;
  for (int round = 0;; ++round) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < MT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (!prefetched) stage(rsA, rsB, 0, 0);
    // DRIP: pending block finished in this K-step (-1: none; -2: none, but the previous K-step's block
    // stores may still be in flight)
    auto kstep = [&](int kt, auto drip_tag) {
      constexpr int DRIP = decltype(drip_tag)::value;
      // K-step kt has landed once only the deferred stores issued AFTER its loads (the previous
      // K-step's block: a fixed number of store instructions per wave) remain outstanding; after the
      // block that ends with the column-sum atomics everything is waited for
      constexpr bool kCounted = (DRIP >= 1 || DRIP == -2) && !(EPI == MVPTR_EPI_GELU_BWD && (DRIP == -2));
      if constexpr (kCounted) {
        asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(STORES_PER_BLOCK) : "memory");
      } else if constexpr (DRIP == 0) {
        // first K-step of a tile: when the previous tile's end requested K-steps 0 AND 1, only K-step 0
        // has to have landed (the LDS-DMA instructions of K-step 1 may stay in flight)
        if (prefetched) asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(NA + NB) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
      } else {
        asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
      }
      if constexpr (DRIP >= 0) drip_load(DRIP);
      const int buf = kt & 1;
      const char* la = lds + buf * STAGE_BYTES;
      const char* lb = la + A_BYTES;
      bf16x8 xf[MT], wf[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) wf[i] = *reinterpret_cast<const bf16x8*>(lb + fw0[0] + i * 16 * ROW_B);
#pragma unroll
      for (int i = 0; i < MT; ++i) xf[i] = *reinterpret_cast<const bf16x8*>(la + fx0[0] + i * 16 * ROW_B);
      if (kt + 1 < nk && !(prefetched && kt == 0)) stage(rsA, rsB, buf ^ 1, (kt + 1) * BK);
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        if (ks > 0) {
#pragma unroll
          for (int i = 0; i < 4; ++i) wf[i] = *reinterpret_cast<const bf16x8*>(lb + fw0[ks] + i * 16 * ROW_B);
#pragma unroll
          for (int i = 0; i < MT; ++i) xf[i] = *reinterpret_cast<const bf16x8*>(la + fx0[ks] + i * 16 * ROW_B);
        }
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
          for (int mt = 0; mt < MT; ++mt)
            acc[nt][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[nt], xf[mt], acc[nt][mt], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
        if constexpr (DRIP >= 0) {
          if (ks == 0) drip(DRIP);
        }
      }
    };
    kstep(0, std::integral_constant<int, 0>{});
    kstep(1, std::integral_constant<int, 1>{});
    kstep(2, std::integral_constant<int, 2>{});
    kstep(3, std::integral_constant<int, 3>{});
    kstep(4, std::integral_constant<int, 4>{});
    kstep(5, std::integral_constant<int, 5>{});
    if (nk > 6) kstep(6, std::integral_constant<int, -2>{});
    for (int kt = 7; kt < nk; ++kt) kstep(kt, std::integral_constant<int, -1>{});
    asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");  // every wave is done with the operand ring
    int m1 = 0, n1 = 0;
    const bool more = tile_origin(round + 1, m1, n1);
    __amdgpu_buffer_rsrc_t rsA1 = rsA, rsB1 = rsB;
    // the tile's bias quads first (their wait must not include the next tile's loads), then the next
    // tile's first two K-steps, then accumulators (+ bias in f32) -> bf16 pending registers
    f32x4 b4[4];
    {
      int lane_e = lane;
      asm volatile("" : "+v"(lane_e));
      const int nb0 = n0 + wn * 64 + (lane_e >> 4) * 4;
      constexpr bool kBias = (EPI != MVPTR_EPI_GELU_BWD && EPI != MVPTR_EPI_ADD);
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) {
        b4[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
        const int n = nb0 + nt * 16;
        if (kBias && p.bias != nullptr && n < p.N) b4[nt] = *reinterpret_cast<const f32x4*>(p.bias + n);  // N % 8 == 0: whole quads
      }
      asm volatile("" : "+v"(b4[0]), "+v"(b4[1]), "+v"(b4[2]), "+v"(b4[3]));   // landed before the loads below are issued
    }
    if (more) {
      rsA1 = rsrc_a(m1);
      rsB1 = rsrc_b(n1);
      stage(rsA1, rsB1, 0, 0);
      stage(rsA1, rsB1, 1, BK);
    }
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        const bf16x2 lo = {f2bf(acc[nt][mt][0] + b4[nt][0]), f2bf(acc[nt][mt][1] + b4[nt][1])};
        const bf16x2 hi = {f2bf(acc[nt][mt][2] + b4[nt][2]), f2bf(acc[nt][mt][3] + b4[nt][3])};
        pend[mt][nt][0] = __builtin_bit_cast(uint32_t, lo);
        pend[mt][nt][1] = __builtin_bit_cast(uint32_t, hi);
      }
    pm0 = m0;
    pn0 = n0;
    if (!more) break;
    m0 = m1;
    n0 = n1;
    rsA = rsA1;
    rsB = rsB1;
    prefetched = true;
  }
  // the last tile's epilogue
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    drip_load(mt);
    drip(mt);
  }
}

template <int EPI>
int launch_pd(GemmNtArgs a, hipStream_t s) {
  constexpr int LDS_BYTES = 2 * (192 + 256) * 64 * 2 + 8 * 16 * 64 * 4;  // 112 KiB ring + 32 KiB staging
  a.tiles_m = (a.M + 191) / 192;
  a.tiles_n = (a.N + 255) / 256;
  if ((int64_t)256 * a.lda * 2 >= (int64_t)0x7fffffff || (int64_t)256 * a.ldb * 2 >= (int64_t)0x7fffffff ||
      (int64_t)a.M * a.ldc * 2 >= (int64_t)0x7fffffff)
    MVPTR_FAIL(MVPTR_BAD_SHAPE, "gemm_nt: leading dimension too large");
  hipError_t e = hipFuncSetAttribute((const void*)gemm_nt_pd_kernel<EPI>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
  if (e != hipSuccess) MVPTR_FAIL(MVPTR_HIP_ERROR, "gemm_nt: set LDS size: %s", hipGetErrorString(e));
  static int num_cu = 0;
  if (num_cu == 0) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess)
      MVPTR_FAIL(MVPTR_HIP_ERROR, "gemm_nt: no HIP device");
    num_cu = prop.multiProcessorCount;
  }
  const int ntiles = a.tiles_m * a.tiles_n;
  const int grid = ntiles < num_cu ? ntiles : num_cu;
  hipLaunchKernelGGL((gemm_nt_pd_kernel<EPI>), dim3(grid), dim3(512), LDS_BYTES, s, a);
  MVPTR_CHECK_LAUNCH("gemm_nt");
  return MVPTR_OK;
}
template <int EPI>
constexpr bool pd_has_epilogue() {
  return EPI == MVPTR_EPI_BIAS || EPI == MVPTR_EPI_BIAS_GELU || EPI == MVPTR_EPI_BIAS_RESID || EPI == MVPTR_EPI_GELU_BWD ||
         EPI == MVPTR_EPI_ADD || EPI == MVPTR_EPI_BIAS_TANH;
}
// whole 8-column runs, 16-byte accesses, six K-steps to drip into
inline bool pd_eligible(const GemmNtArgs& a) {
  if ((a.N & 7) || !a.vec_out_ok || a.K < 384) return false;
  if (a.aux != nullptr && !a.vec_aux_ok) return false;
  if (a.bias != nullptr && !a.vec_bias_ok) return false;
  return true;
}

template <int EPI>
int launch_persist(GemmNtArgs a, hipStream_t s) {
  using C = Cfg<64, 2, 2, 4, 8>;
  constexpr int LDS_BYTES = C::LDS_BYTES + 8 * 16 * 64 * 4;  // ring + 32 KiB staging = 160 KiB
  a.tiles_m = (a.M + C::BM - 1) / C::BM;
  a.tiles_n = (a.N + C::BN - 1) / C::BN;
  if ((int64_t)C::BM * a.lda * 2 >= (int64_t)0x7fffffff || (int64_t)C::BN * a.ldb * 2 >= (int64_t)0x7fffffff)
    MVPTR_FAIL(MVPTR_BAD_SHAPE, "gemm_nt: leading dimension too large");
  hipError_t e = hipFuncSetAttribute((const void*)gemm_nt_persist_kernel<EPI>,
                                     hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
  if (e != hipSuccess) MVPTR_FAIL(MVPTR_HIP_ERROR, "gemm_nt: set LDS size: %s", hipGetErrorString(e));
  static int num_cu = 0;
  if (num_cu == 0) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess)
      MVPTR_FAIL(MVPTR_HIP_ERROR, "gemm_nt: no HIP device");
    num_cu = prop.multiProcessorCount;
  }
  const int ntiles = a.tiles_m * a.tiles_n;
  const int grid = ntiles < num_cu ? ntiles : num_cu;
  hipLaunchKernelGGL((gemm_nt_persist_kernel<EPI>), dim3(grid), dim3(512), LDS_BYTES, s, a);
  MVPTR_CHECK_LAUNCH("gemm_nt");
  return MVPTR_OK;
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
  hipLaunchKernelGGL((gemm_nt_kernel<EPI, BK, STAGES, WM, WN, MT_, SCHED>), dim3(nwg), dim3(WM * WN * 64), LDS_BYTES, s, a);
  MVPTR_CHECK_LAUNCH("gemm_nt");
  return MVPTR_OK;
}

template <int EPI>
int launch(const GemmNtArgs& a, hipStream_t s) {
  // Tile configurations (measured on MI355X, round 1, profiles/r01_gemm_configs.txt):
  //   "t256k" 256x256, BK 64 (whole 128-B lines per row), double buffer, 8 waves of 128x64, one
  //           workgroup per CU: fewest L2->LDS bytes per FLOP (32 B/clk/CU at full MFMA rate, the
  //           measured L2->LDS fill rate is ~30-35), loop rate 1.2-1.4 PF/s
  //   "w4"    256x128, BK 32, 3-stage ring, 8 waves of 64x64, two workgroups per CU: the second
  //           workgroup's loop runs beside the first one's epilogue, lower loop rate (1.0-1.15 PF/s)
  //   "t256" (BK 32 ring) / "t256g" (ring + the two wave halves staggered by one MFMA burst) /
  //           "w4g": tuning knobs, within 5 % of the two above
  // Rule (sweeps of tools/sweep_gemm_cfg.py at the step's shapes): "t256k" everywhere except the
  // short-K, narrow GEMMs whose 256x256 tiles would need more than one round of the 256 CUs
  // (attention-output projection on the big batch), where the epilogue overlap of "w4" wins.
  // MVPTR_GEMM_CFG overrides the choice.
  const char* env = mvptr_knobs().gemm_cfg;
  if (env[0] != 0) {
    const size_t n = strlen(env);
    if (env[0] == 'p' && env[1] == 'd') {
      if constexpr (pd_has_epilogue<EPI>()) {
        if (!pd_eligible(a)) MVPTR_FAIL(MVPTR_BAD_ARG, "gemm_nt: MVPTR_GEMM_CFG=pd needs N %% 8 == 0, K >= 384, 16-byte aligned operands");
        return launch_pd<EPI>(a, s);
      } else {
        MVPTR_FAIL(MVPTR_BAD_ARG, "gemm_nt: MVPTR_GEMM_CFG=pd: epilogue not supported");
      }
    }
    if (env[0] == 'q' && env[1] == 'p') {
      if constexpr (qp_has_epilogue<EPI>()) {
        if (!q_eligible(a) || a.K < 512) MVPTR_FAIL(MVPTR_BAD_ARG, "gemm_nt: MVPTR_GEMM_CFG=qp needs the Q conditions and K >= 512");
        return launch_qp<EPI>(a, s);
      } else {
        MVPTR_FAIL(MVPTR_BAD_ARG, "gemm_nt: MVPTR_GEMM_CFG=qp: epilogue not supported");
      }
    }
    if (env[0] == 'q') {
      if (!q_eligible(a)) MVPTR_FAIL(MVPTR_BAD_ARG, "gemm_nt: MVPTR_GEMM_CFG=q needs N %% 8 == 0, 16-byte aligned operands, no vec_out");
      return launch_q<EPI>(a, s);
    }
    if (env[0] == 'p') return launch_persist<EPI>(a, s);  // "p256": persistent 256x256 / BK 64
    if (env[0] == 's') return launch_bk<EPI, 32, 3, 2, 2, 4, 0>(a, s);  // "s128": 128x128, 4 waves
    if (env[0] == 'h') return launch_bk<EPI, 32, 3, 2, 2, 8, 0>(a, s);  // "h4": 256x128, 4 waves of 128x64, 2 WG/CU
    if (env[0] == 't' && env[n - 1] == 'k') return launch_bk<EPI, 64, 2, 2, 4, 8, 0>(a, s);
    if (env[0] == 't' && env[n - 1] == 'g') return launch_bk<EPI, 32, 3, 2, 4, 8, 2>(a, s);
    if (env[0] == 't') return launch_bk<EPI, 32, 3, 2, 4, 8, 0>(a, s);
    if (env[0] == 'w' && env[n - 1] == 'g') return launch_bk<EPI, 32, 3, 4, 2, 4, 2>(a, s);
    if (env[0] == 'w') return launch_bk<EPI, 32, 3, 4, 2, 4, 0>(a, s);
    MVPTR_FAIL(MVPTR_BAD_ARG, "gemm_nt: unknown MVPTR_GEMM_CFG '%s'", env);
  }
  const int64_t tiles256 = (int64_t)((a.M + 255) / 256) * ((a.N + 255) / 256);
  // few-row GEMMs (head transforms on the masked rows: M ~ 3 k, N = 768) would give a 256x256 tile to
  // a quarter of the CUs or fewer: 128x128 tiles, 4 waves, up to three workgroups per CU ("s128":
  // 35 vs 74 us at M = 3000, N = 768, K = 3072; at M = 11 k the big tile still wins, 74 vs 87 us)
  if (tiles256 <= 64 && a.M > 128) return launch_bk<EPI, 32, 3, 2, 2, 4, 0>(a, s);
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
  a.delay_lo = 256;
  a.delay_hi = 512;
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

extern "C" int mvptr_gemm_nt(const void* A, int64_t lda, const void* B, int64_t ldb, int M, int N,
                             int K, int epilogue, const float* bias, const void* aux,
                             int64_t ld_aux, void* out0, void* out1, int64_t ldc, float* vec_out,
                             const mvptr_dropout* drop, void* stream) {
  if (M <= 0 || N <= 0 || K <= 0) MVPTR_FAIL(MVPTR_BAD_SHAPE, "gemm_nt: M,N,K must be > 0");
  if ((K & 7) || (lda & 7) || (ldb & 7))
    MVPTR_FAIL(MVPTR_BAD_ALIGN, "gemm_nt: K, lda, ldb must be multiples of 8 (K=%d lda=%ld ldb=%ld)",
               K, (long)lda, (long)ldb);
  if (((uintptr_t)A & 15) || ((uintptr_t)B & 15))
    MVPTR_FAIL(MVPTR_BAD_ALIGN, "gemm_nt: A and B must be 16-byte aligned");
  if (lda < K || ldb < K) MVPTR_FAIL(MVPTR_BAD_SHAPE, "gemm_nt: lda/ldb smaller than K");
  if (out0 == nullptr) MVPTR_FAIL(MVPTR_BAD_ARG, "gemm_nt: out0 is NULL");
  GemmNtArgs a;
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
  a.out0 = out0;
  a.out1 = out1;
  a.ldc = ldc;
  a.vec_out = vec_out;
  a.drop = make_dropdev(drop);
  a.stamps = nullptr;
  a.delay_cycles = 0;
  const MvptrKnobs& kn = mvptr_knobs();
  a.exp_flags = kn.nt_exp;
  a.delay_cycles = kn.delay[0];
  a.delay_lo = kn.delay[1];
  a.delay_hi = kn.delay[2];
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
    default:
      MVPTR_FAIL(MVPTR_BAD_ARG, "gemm_nt: unknown epilogue %d", epilogue);
  }
}
