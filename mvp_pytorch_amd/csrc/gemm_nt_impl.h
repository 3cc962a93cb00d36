// gemm_nt_impl.h — what the forward / data-gradient GEMM kernels of gemm_nt.hip and the measured-and-rejected experiment
// kernels of diag_gemm.hip (diagnostic build only) share: launch arguments, tile geometry, the epilogue.
#pragma once
#include "common.h"
#include <stdlib.h>
#include <string.h>
#include <type_traits>

// launch arguments (external linkage: gemm_nt.hip hands them to diag_gemm.hip in the diagnostic build)
struct GemmNtArgs {
  const __bf16* A;
  const __bf16* B;
  int64_t lda, ldb;
  int M, N, K;
  const int* rows_dev;   // device int32: actual row count <= M (NULL: M); workgroups beyond it return at once
  int m_plan;            // rows the tile configuration is chosen for (<= M; the grid always covers M)
  int full_height;       // 1: keep 256-row tiles (another stream fills under-filled CU rounds: mvptr_layer_desc.beside)
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
  int epi_ablate;      // diagnostic build (MVPTR_NT_EXP bits 26-30): 1 = aux rows not loaded, 2 = outputs not stored, 4 = no column-sum atomics,
                       //   8 = GELU arithmetic skipped (identity), 16 = the A&S GELU of rounds 1-5 — where does an epilogue's time go
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
  // LayerNorm folded into the neighbouring GEMMs of the inference path (mvptr_gemm_nt_ln, EPI_FOLD_* / EPI_RESID_LN below)
  const float* ln_stats;  // [M, 2] (mean, rstd) of the rows of the pre-LayerNorm operand (A for EPI_FOLD_*, aux for EPI_RESID_LN; NULL: aux is final)
  const float* ln_c;      // [N] EPI_FOLD_*: column sums of the gamma-scaled weight, c[n] = sum_k W'[n, k]
  const float* ln_gamma;  // [N] EPI_RESID_LN: weight / bias of the LayerNorm applied to the residual rows on the fly
  const float* ln_beta;
};

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

// library-internal epilogues of the fused decoder + cross-entropy entry points
constexpr int EPI_CE_PART = 16;  // per row and 64-column wave strip: (max, sum exp(v - max)) of v = acc + bias; logit at the label
constexpr int EPI_CE_BWD = 17;   // out0(bf16) = (exp(v - lse[m]) - [n == label[m]]) * scale, 0 for unscored rows / pad columns
// LayerNorm folded into its consumer (inference path; north_star "fused LayerNorm + QKV projection").  With x = LN(z) =
// (z - mu) rstd gamma + beta per row:  x W^T + b = rstd (z W'^T - mu c) + d,  W' = gamma o W (columns scaled), c = rowsum(W'),
// d = W beta + b.  The GEMM runs on the PRE-LayerNorm rows z with W'; the epilogue applies the row statistics:
constexpr int EPI_FOLD_BIAS = 18;  // out0(bf16) = rstd[m] (acc - mean[m] c[n]) + d[n]                       (Q/K/V projection)
constexpr int EPI_FOLD_GELU = 19;  // out0(bf16) = gelu(the same)                                           (FFN1; no gelu' stash: inference)
// ... and its producer: the GEMM in front of a LayerNorm writes the pre-LayerNorm rows and their statistics' partial sums,
constexpr int EPI_RESID_LN = 20;   // z = acc + bias + r, r = aux (ln_stats NULL) or LN(aux) from ln_stats / ln_gamma / ln_beta; out0(bf16) = z;
                                   // part[m, strip, 0..1] = (sum, sum of squares) of the ROUNDED z over the 64 columns of wave strip n / 64

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
#ifdef MVPTR_DIAG_BUILD
  const int abl = p.epi_ablate;
#else
  constexpr int abl = 0;
#endif

  const DropDev drop = (EPI == MVPTR_EPI_BIAS_RESID) ? drop_resolve(p.drop) : p.drop;
  const int ch = lane & 7, rsub = lane >> 3;
  const int n = n0 + wn * 64 + ch * 8;
  const bool nfull = (n + 7 < p.N);
  float b8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (EPI != MVPTR_EPI_GELU_BWD && EPI != MVPTR_EPI_GELU_BWD_BF16 && EPI != MVPTR_EPI_ADD && p.bias != nullptr && n < p.N) {
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
    if (abl & 2) return;
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

  constexpr bool kNeedsAux = (EPI == MVPTR_EPI_BIAS_RESID || EPI == MVPTR_EPI_GELU_BWD || EPI == MVPTR_EPI_GELU_BWD_BF16 || EPI == MVPTR_EPI_ADD ||
                              EPI == EPI_RESID_LN);
  constexpr bool kFold = (EPI == EPI_FOLD_BIAS || EPI == EPI_FOLD_GELU);
  // per-column vectors of the folded LayerNorm: c (EPI_FOLD_*), gamma / beta of the residual's LayerNorm (EPI_RESID_LN)
  float lc8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, lb8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if constexpr (kFold || EPI == EPI_RESID_LN) {
    const float* cv = kFold ? p.ln_c : p.ln_gamma;
    if (cv != nullptr) {
#pragma unroll
      for (int e = 0; e < 8; ++e)
        if (n + e < p.N) {
          lc8[e] = cv[n + e];
          if constexpr (EPI == EPI_RESID_LN) lb8[e] = p.ln_beta[n + e];
        }
    }
  }
  const bool has_aux = kNeedsAux && p.aux != nullptr && !(abl & 1);

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
            if (2 * ck + mh < MT)       // odd MT: the last 32-row chunk holds one 16-row block
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
    if constexpr (kFold) {
      const f32x2 st = *reinterpret_cast<const f32x2*>(p.ln_stats + 2 * (int64_t)m);      // (mean, rstd) of row m
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        v[e] = st.y * (v0[e] - st.x * lc8[e]) + b8[e];
        v[4 + e] = st.y * (v1[e] - st.x * lc8[4 + e]) + b8[4 + e];
      }
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
    if (EPI == MVPTR_EPI_BIAS || EPI == EPI_FOLD_BIAS) {
      store_bf8(p.out0, m, v);
    } else if (EPI == EPI_FOLD_GELU) {
      float g[8], dg[8];
      gelu8(v, g, dg);
      store_bf8(p.out0, m, g);
    } else if (EPI == EPI_RESID_LN) {
      if (p.ln_stats != nullptr) {      // the residual rows are pre-LayerNorm rows: normalise them here
        const f32x2 st = *reinterpret_cast<const f32x2*>(p.ln_stats + 2 * (int64_t)m);
#pragma unroll
        for (int e = 0; e < 8; ++e) a[e] = (a[e] - st.x) * st.y * lc8[e] + lb8[e];
      }
      float s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        v[e] = bf2f(f2bf(v[e] + a[e]));      // the statistics are those of the rows as stored (what a LayerNorm kernel would read)
        if (n + e < p.N) {
          s1 += v[e];
          s2 += v[e] * v[e];
        }
      }
      store_bf8(p.out0, m, v);
      // the 8 lanes that share the row (lane bits 0-2) hold the wave strip's 64 columns
      s1 = MVPTR_DPP_ADD(s1, 0xB1);
      s2 = MVPTR_DPP_ADD(s2, 0xB1);
      s1 = MVPTR_DPP_ADD(s1, 0x4E);
      s2 = MVPTR_DPP_ADD(s2, 0x4E);
      s1 = MVPTR_DPP_ADD(s1, 0x141);
      s2 = MVPTR_DPP_ADD(s2, 0x141);
      if (ch == 0) *reinterpret_cast<f32x2*>(p.part + ((int64_t)m * p.part_ld + ((n0 >> 6) + wn)) * 2) = f32x2{s1, s2};
    } else if (EPI == MVPTR_EPI_BIAS_GELU) {
      float g[8], dg[8];
#ifdef MVPTR_DIAG_BUILD
      if (abl & 24) {
#pragma unroll
        for (int e = 0; e < 8; e += 2) {
          f32x2 a2, d2;
          if (abl & 8) a2 = d2 = f32x2{v[e], v[e + 1]};
          else gelu_pair_as(f32x2{v[e], v[e + 1]}, a2, d2);
          g[e] = a2.x;
          g[e + 1] = a2.y;
          dg[e] = d2.x;
          dg[e + 1] = d2.y;
        }
      } else
#endif
      gelu8(v, g, dg);
      // gelu'(u) is only read in the backward pass: non-temporal stores keep it from displacing gelu(u) — the next
      // GEMM's operand — in the Infinity Cache (same-box A/B: all-slots step 41.71 -> 41.46 ms, packed 28.68 -> 28.60)
      // out0 = the 8-bit gelu' stash (common.h), one byte per element, row stride ldc bytes
      uint8_t* dp = reinterpret_cast<uint8_t*>(p.out0) + (int64_t)m * p.ldc + n;
      const u32x2 dq = {dgelu_pack4_dither(dg[0], dg[1], dg[2], dg[3], v[0], v[1], v[2], v[3]),
                        dgelu_pack4_dither(dg[4], dg[5], dg[6], dg[7], v[4], v[5], v[6], v[7])};
      if (abl & 2) {
      } else if (nfull && p.vec_out_ok) {
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
    } else if (EPI == MVPTR_EPI_BIAS_GELU_BF16) {
      float g[8], dg[8];
      gelu8(v, g, dg);
      store_bf8(p.out0, m, dg);      // the bf16 stash of rounds 1-3
      store_bf8(p.out1, m, g);
    } else if (EPI == MVPTR_EPI_BIAS_RESID) {
      if (drop.thresh16 != 0) {
        if ((p.N & 1) == 0) {   // (m * N + n) even: lanes own whole hash pairs; the four pairs of the row piece hashed in lockstep
          uint64_t pr[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) pr[e] = (((uint64_t)m * (uint64_t)p.N + (uint64_t)n) >> 1) + (uint64_t)e;
          // every element index fits 32 bits (all of this model's outputs): 32-bit index arithmetic, two multiplies per hash
          if ((uint64_t)p.M * (uint64_t)p.N < ((uint64_t)1 << 32)) drop_pairs<4, true>(drop, pr, v);
          else drop_pairs<4, false>(drop, pr, v);
        } else {
#pragma unroll
          for (int e = 0; e < 8; ++e)
            v[e] = drop_apply(drop, (uint64_t)m * (uint64_t)p.N + (uint64_t)(n + e), v[e]);
        }
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] += a[e];
      store_bf8(p.out0, m, v);
    } else if (EPI == MVPTR_EPI_GELU_BWD || EPI == MVPTR_EPI_GELU_BWD_BF16) {
      // aux = gelu'(u) saved by the forward epilogue (8-bit stash or bf16, decoded above); rows / columns outside the problem have
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
  if ((EPI == MVPTR_EPI_GELU_BWD || EPI == MVPTR_EPI_GELU_BWD_BF16) && p.vec_out != nullptr && !(abl & 4)) {
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

}  // namespace

#ifdef MVPTR_DIAG_BUILD
// diag_gemm.hip: the experiment kernels behind MVPTR_GEMM_CFG = n768 | p | pd.  Returns MVPTR_DIAG_NOT_HANDLED when the
// configuration does not take this epilogue / shape (the caller then launches the product kernel).
constexpr int MVPTR_DIAG_NOT_HANDLED = 1;
int mvptr_diag_gemm_nt(int epilogue, const GemmNtArgs& a, const char* cfg, hipStream_t s);
#endif
