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
// time of a 192-row tile relative to three quarters of a 256-row tile's (launch(): tile height)
constexpr double kShortTilePenalty = 1.1;
// chunk swizzles that make the 16x16x32 ds_read_b128 fragment reads conflict free
__device__ __forceinline__ int swz_row(int row, int chunks) {
  return chunks == 8 ? ((row >> 1) & 7) : ((0x78 >> (((row >> 2) & 3) * 2)) & 3);  // LUT {0,2,3,1}
}

int nt_num_cus() {
  static int n = 0;
  if (n == 0) {
    int dev = 0, v = 0;
    if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) n = v;
    else n = 256;
  }
  return n;
}

struct GemmNtArgs {
  const __bf16* A;
  const __bf16* B;
  int64_t lda, ldb;
  int M, N, K;
  const int* rows_dev;   // device int32: actual row count <= M (NULL: M); workgroups beyond it return at once
  int m_plan;            // rows the tile configuration is chosen for (<= M; the grid always covers M)
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
  int splits;          // gridDim.y (1 unless split-K)
  int k_split_len;     // split-K launches (mvptr_gemm_nt_splitk): workgroups with blockIdx.y = z reduce over k in [z * k_split_len, +k_split_len)
  int64_t slab_stride; //   and write their f32 partial tile into slab z = out0 + z * slab_stride elements; 0 = whole K, no slabs
  int no_epi;          // diagnostic build (MVPTR_NT_EXP bit 10): skip the epilogue (loop-only timing; outputs are not written)
  int store_mode;      // diagnostic build (MVPTR_NT_EXP bits 13-15, persistent kernel): 1 = stores dropped (zero-size descriptor), 2 = nt, 3 = sc1, 4 = sc0 sc1
  int stash_temporal;  // diagnostic build (MVPTR_NT_EXP bit 9): EPI_BIAS_GELU stores gelu'(u) with plain instead of non-temporal stores (A/B)
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
__device__ __forceinline__ void nt_epilogue(const GemmNtArgs& p, int Mv, f32x4 (&acc)[4][MT], float* st, int m0, int n0,
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
      *reinterpret_cast<bf16x8*>(op) = o;
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e)
        if (n + e < p.N) op[e] = f2bf(v[e]);
    }
  };

  constexpr bool kNeedsAux = (EPI == MVPTR_EPI_BIAS_RESID || EPI == MVPTR_EPI_GELU_BWD || EPI == MVPTR_EPI_ADD);
  const bool has_aux = kNeedsAux && p.aux != nullptr;

  // residual / pre-activation rows (aux): the four rows of a 32-row chunk are requested together, ONE CHUNK AHEAD of the
  // chunk being finished — issued in front of that chunk's stores, so they are older in the wave's in-order vmcnt queue
  // and their latency (HBM: the operand was written kernels ago) runs under the chunk's LDS round trip, math and stores
  // raw bits: eight bf16 (16 bytes), or for EPI_GELU_BWD the eight bytes of the 8-bit gelu' stash (words 0, 1)
  u32x4 auxv[2][4];
  auto load_aux = [&](int it0, u32x4 (&dstv)[4]) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int mj = m0 + wm * WROWS + (it0 + j) * 8 + rsub;
      u32x4 x = {0u, 0u, 0u, 0u};
      if (has_aux && mj < Mv && n < p.N) {
        if constexpr (EPI == MVPTR_EPI_GELU_BWD) {
          const uint8_t* ap = reinterpret_cast<const uint8_t*>(p.aux) + (int64_t)mj * p.ld_aux + n;
          if (nfull && p.vec_aux_ok) {
            const u32x2 w = *reinterpret_cast<const u32x2*>(ap);
            x[0] = w[0];
            x[1] = w[1];
          } else {
#pragma unroll
            for (int e = 0; e < 8; ++e)
              if (n + e < p.N) x[e >> 2] |= (uint32_t)ap[e] << (8 * (e & 3));
          }
        } else {
          const __bf16* ap = p.aux + (int64_t)mj * p.ld_aux + n;
          if (nfull && p.vec_aux_ok) {
            x = *reinterpret_cast<const u32x4*>(ap);
          } else {
            bf16x8 t;
#pragma unroll
            for (int e = 0; e < 8; ++e) t[e] = (n + e < p.N) ? ap[e] : f2bf(0.f);
            x = __builtin_bit_cast(u32x4, t);
          }
        }
      }
      dstv[j] = x;
    }
  };
  if (kNeedsAux) load_aux(0, auxv[0]);

#pragma unroll
  for (int it = 0; it < MT * 2; ++it) {
    // 32-row chunk ck = it >> 2 of the wave's block goes through the staging area
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
    if (kNeedsAux && (it & 3) == 0 && it + 4 < MT * 2) load_aux(it + 4, auxv[((it >> 2) + 1) & 1]);
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
      if (m < Mv) {
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
    if (m >= Mv || n >= (EPI == EPI_CE_BWD ? p.n_store : p.N)) continue;
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
      const u32x4 aw = auxv[(it >> 2) & 1][it & 3];
      if constexpr (EPI == MVPTR_EPI_GELU_BWD) {
#pragma unroll
        for (int e = 0; e < 8; ++e) a[e] = dgelu_unpack(aw[e >> 2], e & 3);
      } else {
        const bf16x8 ab = __builtin_bit_cast(bf16x8, aw);
#pragma unroll
        for (int e = 0; e < 8; ++e) a[e] = bf2f(ab[e]);
      }
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
      // gelu'(u) is only read in the backward pass: non-temporal stores keep it from displacing gelu(u) — the next
      // GEMM's operand — in the Infinity Cache (same-box A/B: all-slots step 41.71 -> 41.46 ms, packed 28.68 -> 28.60)
      // out0 = the 8-bit gelu' stash (common.h), one byte per element, row stride ldc bytes
      uint8_t* dp = reinterpret_cast<uint8_t*>(p.out0) + (int64_t)m * p.ldc + n;
      const u32x2 dq = {dgelu_pack4(dg[0], dg[1], dg[2], dg[3]), dgelu_pack4(dg[4], dg[5], dg[6], dg[7])};
      if (nfull && p.vec_out_ok) {
#ifdef MVPTR_DIAG_BUILD
        if (p.stash_temporal >= 2) {
          // A/B of the cache policy of these half-line (64 bytes per row and wave) stores: 2 = sc1, 3 = sc0 sc1, 4 = nt through
          // the same buffer-store path (MVPTR_NT_EXP bits 19-21)
          const __amdgpu_buffer_rsrc_t rs = make_rsrc(p.out0, 0xfffffff0u);          // one descriptor, per-lane byte offsets
          const uint32_t vo = (uint32_t)((int64_t)m * p.ldc + n);
          const int aux = p.stash_temporal == 2 ? 16 : (p.stash_temporal == 3 ? 17 : 2);
          if (aux == 16) __builtin_amdgcn_raw_buffer_store_b64(dq, rs, vo, 0, 16);
          else if (aux == 17) __builtin_amdgcn_raw_buffer_store_b64(dq, rs, vo, 0, 17);
          else __builtin_amdgcn_raw_buffer_store_b64(dq, rs, vo, 0, 2);
        } else
#endif
        if (!p.stash_temporal) __builtin_nontemporal_store(dq, reinterpret_cast<u32x2*>(dp));
        else *reinterpret_cast<u32x2*>(dp) = dq;
      } else {
#pragma unroll
        for (int e = 0; e < 8; ++e)
          if (n + e < p.N) dp[e] = (uint8_t)(dq[e >> 2] >> (8 * (e & 3)));
      }
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
      // aux = gelu'(u) saved by the forward epilogue (8-bit stash, decoded above); rows / columns outside the problem have
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
      float* op = (float*)p.out0 + (int64_t)blockIdx.y * p.slab_stride + (int64_t)m * p.ldc + n;   // split-K: slab blockIdx.y
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

#ifdef MVPTR_DIAG_BUILD
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
#endif

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
#ifdef MVPTR_DIAG_BUILD
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
            for (int e = 0; e < 8; e += 2) drop_apply2(p.drop, di + (uint64_t)e, v[e], v[e + 1]);
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
#endif  // MVPTR_DIAG_BUILD (persistent ring experiment)

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
    if (env[0] == 'n') {                                                  // "n768": row-owning tile experiment
      if constexpr (EPI == MVPTR_EPI_BIAS_RESID || EPI == MVPTR_EPI_BIAS || EPI == MVPTR_EPI_ADD) return launch_rowtile(a, s);
      else MVPTR_FAIL(MVPTR_BAD_ARG, "gemm_nt: MVPTR_GEMM_CFG=n768 supports the bias / residual / add epilogues only");
    }
    if (env[0] == 'p') {                                                  // "p": persistent ring experiment
      if constexpr (EPI == MVPTR_EPI_BIAS || EPI == MVPTR_EPI_BIAS_GELU || EPI == MVPTR_EPI_BIAS_RESID || EPI == MVPTR_EPI_GELU_BWD ||
                    EPI == MVPTR_EPI_ADD) {
        if (ntp_eligible<EPI>(a)) return (env[1] == 'd' && a.K >= 32 * 24) ? launch_ntp<EPI, true>(a, s) : launch_ntp<EPI, false>(a, s);   // "pd": deferred stores
      }
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

extern "C" int mvptr_gemm_nt(const void* A, int64_t lda, const void* B, int64_t ldb, int M, int N,
                             int K, int epilogue, const float* bias, const void* aux,
                             int64_t ld_aux, void* out0, void* out1, int64_t ldc, float* vec_out,
                             const mvptr_dropout* drop, void* stream) {
  return mvptr_gemm_nt_rows(A, lda, B, ldb, M, N, K, epilogue, bias, aux, ld_aux, out0, out1, ldc, vec_out, drop, nullptr, 0, stream);
}

int mvptr_gemm_nt_rows(const void* A, int64_t lda, const void* B, int64_t ldb, int M, int N, int K, int epilogue, const float* bias,
                       const void* aux, int64_t ld_aux, void* out0, void* out1, int64_t ldc, float* vec_out, const mvptr_dropout* drop,
                       const int* rows_dev, int M_plan, void* stream) {
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
  const MvptrKnobs& kn = mvptr_knobs();
  a.stash_temporal = (kn.nt_exp & 512) ? 1 : 0;
  if ((kn.nt_exp >> 19) & 7) a.stash_temporal = 1 + ((kn.nt_exp >> 19) & 7);      // bits 19-21: 1 = sc1, 2 = sc0 sc1, 3 = nt (buffer store)
  a.no_epi = (kn.nt_exp & 1024) ? 1 : 0;
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
