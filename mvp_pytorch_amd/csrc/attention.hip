// attention.hip — joint text+region self-attention core, forward and backward.
//
// Replaces oscar/modeling/modeling_vlbert.py:75-100 (CaptionBertSelfAttention.forward after
// the Q/K/V projections: transpose_for_scores, QK^T / sqrt(64) + additive mask, softmax,
// dropout, P·V, head merge) and its autograd backward.
//
// CDNA4 design: the sequences on this path are short (<= 70 tokens + <= 50 regions; <= 256
// supported), so one 256-thread workgroup owns one (batch, head) and keeps the whole Q, K, V
// (and dO in backward) [L,64] bf16 tiles in LDS (buffer_load ... lds, 128-B rows, one XOR
// swizzle that is conflict free for both ds_read_b128 row reads and ds_read_b64_tr_b16
// transposed reads).  All products are v_mfma_f32_32x32x16_bf16.  Scores are computed
// transposed (S^T = K·Q^T) so a lane owns one query column: the softmax reductions are
// in-register plus one lane^32 exchange, and the probability accumulator feeds the next MFMA
// (O^T = V^T·P^T) directly as its B operand with no LDS round trip.  Nothing of size L x L
// ever reaches HBM; backward recomputes P from Q, K and the saved row log-sum-exp:
//   pass A (key on lane):   dV^T += dO^T·P,  dK^T += Q^T·dS      (waves own key blocks)
//   pass B (query on lane): dQ^T += K^T·dS^T                     (waves own query blocks)
#include "common.h"

namespace {

struct AttnArgs {
  const __bf16* qkv;
  const float* mask;
  __bf16* ctx;        // fwd: out ; bwd: forward output (for delta)
  const __bf16* dctx; // bwd
  float* lse;         // fwd: out (may be null) ; bwd: in
  __bf16* dqkv;       // bwd out
  int B, L, heads, Lp;      // L / Lp: (maximum) sequence length and its multiple-of-32 tile rows
  const int* seq_start;     // packed mode: first row of sequence b in the row-packed buffers
  const int* seq_len;       //              its length (<= L); NULL => dense [B, L] layout
  DropDev drop;
  unsigned long long* stamps;  // -DMVPTR_TIMELINE_BUILD only (MVPTR_GEMM_STAMPS): per-workgroup phase times
};

// swizzle of the 8 16-byte chunks of a 128-B tile row
__device__ __forceinline__ int swz128(int row) {
  return (((row >> 1) & 1) << 2) | (((row >> 2) & 1) << 1) | ((row >> 3) & 1);
}
__device__ __forceinline__ uint32_t tile_off(int row, int chunk) {
  return (uint32_t)(row * 128 + ((chunk ^ swz128(row)) << 4));
}

// stage an [L,64] bf16 head slice (row stride ld elements) into an Lp-row LDS tile
__device__ __forceinline__ void stage_tile(char* tile, const __bf16* src, int64_t ld, int L, int Lp,
                                           int wave, int lane) {
  const __amdgpu_buffer_rsrc_t rs = make_rsrc(src, (uint32_t)(((int64_t)(L - 1) * ld + 64) * 2));
  const int ninstr = Lp >> 3;  // 8 rows per wave instruction
  for (int i = wave; i < ninstr; i += 4) {
    const int row = i * 8 + (lane >> 3);
    const int c = (lane & 7) ^ swz128(row);
    const uint32_t off = (uint32_t)(row * ld * 2 + c * 16);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, LDS_PTR(tile + i * 1024), 16, off, 0, 0, 0);
  }
}

// The swizzle only looks at row bits 1..3, so for the two access shapes of these kernels the
// swizzled part of the address is a per-lane constant and the block index adds a multiple of
// 4 KiB (32 rows): offsets are computed once per lane instead of per fragment.
struct LaneOffs {
  uint32_t rowf[4];     // row_frag(32*blk + (lane&31), 2*ks + (lane>>5)) - blk*4096, ks = 0..3
  uint32_t trf[2][2];   // tr_frag row0 = 32*blk + 16*st + 4*(lane>>5): [db][lo/hi], minus (32*blk+16*st)*128
};
__device__ __forceinline__ LaneOffs make_lane_offs(int lane) {
  LaneOffs o;
  const int l31 = lane & 31, hh = lane >> 5;
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) o.rowf[ks] = tile_off(l31, 2 * ks + hh);
  const int i16 = lane & 15, cb = (lane >> 4) & 1;
  const int q = i16 >> 2, pp = i16 & 3;
#pragma unroll
  for (int db = 0; db < 2; ++db) {
    const int ch = db * 4 + cb * 2 + (pp >> 1);
    o.trf[db][0] = tile_off(4 * hh + q, ch) + 8 * (pp & 1);
    o.trf[db][1] = tile_off(4 * hh + 8 + q, ch) + 8 * (pp & 1);
  }
  return o;
}
__device__ __forceinline__ bf16x8 row_frag_o(const char* tile, const LaneOffs& o, int blk, int ks) {
  return *reinterpret_cast<const bf16x8*>(tile + blk * 4096 + o.rowf[ks]);
}
// blk16 = 2*blk + st (16-row granularity)
__device__ __forceinline__ bf16x8 tr_frag_o(const char* tile, const LaneOffs& o, int blk16, int db) {
  const char* base = tile + blk16 * 2048;
  s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
      (__attribute__((address_space(3))) s16x4*)LDS_PTR(base + o.trf[db][0]));
  s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
      (__attribute__((address_space(3))) s16x4*)LDS_PTR(base + o.trf[db][1]));
  s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(bf16x8, v);
}
// 16 per-row f32 values of a 32-row block in accumulator-register order (4 x ds_read_b128)
__device__ __forceinline__ void load_rows16(const float* v, int blk, int hh, float out[16]) {
#pragma unroll
  for (int t4 = 0; t4 < 4; ++t4) {
    const f32x4 x = *reinterpret_cast<const f32x4*>(v + 32 * blk + 8 * t4 + 4 * hh);
#pragma unroll
    for (int e = 0; e < 4; ++e) out[4 * t4 + e] = x[e];
  }
}

__device__ __forceinline__ bf16x8 pack8(const f32x16& x, int s) {
  bf16x8 r;
#pragma unroll
  for (int j = 0; j < 8; ++j) r[j] = f2bf(x[8 * s + j]);
  return r;
}

__device__ __forceinline__ f32x16 zero16() {
  f32x16 z;
#pragma unroll
  for (int r = 0; r < 16; ++r) z[r] = 0.f;
  return z;
}

// row index held in accumulator register r by lane half hh (32x32 C/D layout)
__device__ __forceinline__ int acc_row(int r, int hh) { return (r & 3) + 8 * (r >> 2) + 4 * hh; }

// Output rows through LDS (round 4).  The accumulators hold an output TRANSPOSED (row on the lane, 4 consecutive
// columns per register quad), so a direct store instruction wrote 8 bytes per lane = 16 bytes of each of 32 different
// 128-byte lines, eight instructions per line; measured on the packed joint stack those partial-line stores were 56 of
// the backward kernel's 196 us (profiles/r04_experiments.txt: the CU's vector memory path is paid per line touched).
// put_block32 writes a 32-row x 64-column f32 accumulator pair as bf16 into a 4-KiB tile block (rows = the lane's row,
// the tiles' own chunk swizzle), store_block32 reads it back 16 bytes per lane, 8 lanes per row, and stores whole
// lines: 4 instructions per block instead of 8, each touching 8 lines instead of 32.  `blk` must be private to the wave.
__device__ __forceinline__ void put_block32(char* blk, const f32x16 (&acc)[2], float scale, int l31, int hh) {
#pragma unroll
  for (int db = 0; db < 2; ++db)
#pragma unroll
    for (int t4 = 0; t4 < 4; ++t4) {
      const bf16x4 v = {f2bf(acc[db][4 * t4] * scale), f2bf(acc[db][4 * t4 + 1] * scale),
                        f2bf(acc[db][4 * t4 + 2] * scale), f2bf(acc[db][4 * t4 + 3] * scale)};
      *reinterpret_cast<bf16x4*>(blk + tile_off(l31, 4 * db + t4) + 8 * hh) = v;
    }
}
// dst: row 0 of the block in global memory, column 0 of the head; rows_left: rows of the block inside the sequence
__device__ __forceinline__ void store_block32(const char* blk, __bf16* dst, int64_t ld, int rows_left, int lane) {
  const int c = lane & 7;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = 8 * i + (lane >> 3);
    const bf16x8 v = *reinterpret_cast<const bf16x8*>(blk + tile_off(row, c));
    if (row < rows_left) *reinterpret_cast<bf16x8*>(dst + (int64_t)row * ld + 8 * c) = v;
  }
}

extern __shared__ __attribute__((aligned(16))) char smem[];

// ------------------------------------------------------------------------------------ forward
// Dropout of the 16 probabilities a lane holds of a 32 x 32 block (registers r, r + 1 = adjacent keys: 8 hash pairs at
// rowbase + (r & 3) + 8 (r >> 2)), all eight pair hashes in lockstep (common.h pair_hashes: rounds 1-5 hashed one pair after the
// other, three integer multiplies each — the 64-bit form, although every pair index of this model is below 2^32 — and spent more
// issue slots on the masks than on the exponentials).  lo32: wave-uniform, B heads L Lp < 2^33.
template <bool LO32, int G = 4>
__device__ __forceinline__ void drop_block16(const DropDev& d, uint64_t rowbase, f32x16& v) {
  // groups of G pairs (G = 4: eight chains in flight cost registers the kernels do not have)
#pragma unroll
  for (int g = 0; g < 8 / G; ++g) {
    uint32_t h[G];
    if constexpr (LO32) {
      uint32_t pr[G];
#pragma unroll
      for (int j = 0; j < G; ++j) {
        const int r = 2 * (G * g + j);
        pr[j] = ((uint32_t)rowbase + (uint32_t)((r & 3) + 8 * (r >> 2))) >> 1;
      }
      pair_hashes_lo<G>(pr, d.seed_lo, d.seed_hi, h);
    } else {
      uint64_t pr[G];
#pragma unroll
      for (int j = 0; j < G; ++j) {
        const int r = 2 * (G * g + j);
        pr[j] = (rowbase + (uint64_t)((r & 3) + 8 * (r >> 2))) >> 1;
      }
      pair_hashes<G, false>(pr, d.seed_lo, d.seed_hi, h);
    }
#pragma unroll
    for (int j = 0; j < G; ++j) {
      const int r = 2 * (G * g + j);
      v[r] = ((h[j] & 0xffffu) >= d.thresh16) ? v[r] * d.scale : 0.f;
      v[r + 1] = ((h[j] >> 16) >= d.thresh16) ? v[r + 1] * d.scale : 0.f;
    }
  }
}
// Keep bits of the 16 (query register) x (this lane's key) probabilities of a block in the key-on-lane passes: the pair (key & ~1,
// key | 1) of query q shares one hash; this lane hashes the queries whose register parity equals its key parity and swaps with
// lane ^ 1 (DPP) — eight hashes per lane and block, G at a time in lockstep (the three-workgroups-per-CU kernel has 168 registers:
// one at a time there, as rounds 3-5 did — with four in flight its spills cost more than the chains saved: 169 -> 198 us)
template <bool LO32, int G = 4>
__device__ __forceinline__ uint32_t keep_bits16(const DropDev& d, int bh, int LT, int LpT, int qb, int hh, int key) {
  const int par = key & 1;
  uint32_t keepbits = 0;
#pragma unroll
  for (int g = 0; g < 8 / G; ++g) {
    uint32_t h[G];
    if constexpr (LO32) {
      uint32_t pr[G];
#pragma unroll
      for (int j = 0; j < G; ++j) {
        const int rm = 2 * (G * g + j) + par;      // the query register this lane hashes
        const int qm = 32 * qb + (rm & 3) + 8 * (rm >> 2) + 4 * hh;
        pr[j] = (((uint32_t)bh * (uint32_t)LT + (uint32_t)qm) * (uint32_t)LpT + (uint32_t)(key & ~1)) >> 1;
      }
      pair_hashes_lo<G>(pr, d.seed_lo, d.seed_hi, h);
    } else {
      uint64_t pr[G];
#pragma unroll
      for (int j = 0; j < G; ++j) {
        const int rm = 2 * (G * g + j) + par;
        const int qm = 32 * qb + (rm & 3) + 8 * (rm >> 2) + 4 * hh;
        pr[j] = (((uint64_t)bh * LT + qm) * (uint64_t)LpT + (uint64_t)(key & ~1)) >> 1;
      }
      pair_hashes<G, false>(pr, d.seed_lo, d.seed_hi, h);
    }
#pragma unroll
    for (int j = 0; j < G; ++j) {
      const int rm = 2 * (G * g + j) + par;
      const uint32_t hm = h[j];
      const uint32_t ho = (uint32_t)__builtin_amdgcn_mov_dpp((int)hm, 0xB1, 0xF, 0xF, true);  // quad_perm [1,0,3,2]
      const uint32_t um = par ? (hm >> 16) : (hm & 0xffffu);
      const uint32_t uo = par ? (ho >> 16) : (ho & 0xffffu);
      keepbits |= (um >= d.thresh16 ? 1u : 0u) << rm;
      keepbits |= (uo >= d.thresh16 ? 1u : 0u) << (rm ^ 1);
    }
  }
  return keepbits;
}

template <bool LO32>
__global__ __launch_bounds__(256) void attn_fwd_kernel(AttnArgs p) {
  p.drop = drop_resolve(p.drop);      // device-side salt of graph-replayed steps (common.h)
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int bh = blockIdx.x;
  const int b = bh / p.heads, hd = bh - b * p.heads;
  // LT / LpT: table strides (lse, dropout index, LDS tile placement); L / Lp: this sequence
  const int LT = p.L, LpT = p.Lp, H = p.heads * 64;
  constexpr bool lo32 = LO32;      // every dropout element index fits 32 bits (chosen by the launch: B heads L Lp < 2^32)
  const int L = (p.seq_len != nullptr) ? p.seq_len[b] : LT;
  if (L <= 0) return;
  const int64_t row0 = (p.seq_start != nullptr) ? (int64_t)p.seq_start[b] : (int64_t)b * LT;
  const int Lp = (L + 31) & ~31;
  const int64_t ldq = 3 * (int64_t)H;
  char* tQ = smem;
  char* tK = tQ + LpT * 128;
  char* tV = tK + LpT * 128;
  float* maskv = reinterpret_cast<float*>(tV + LpT * 128);

  const __bf16* base = p.qkv + row0 * ldq + hd * 64;
  stage_tile(tQ, base, ldq, L, Lp, wave, lane);
  stage_tile(tK, base + H, ldq, L, Lp, wave, lane);
  stage_tile(tV, base + 2 * H, ldq, L, Lp, wave, lane);
  for (int i = tid; i < Lp; i += 256)
    maskv[i] = (i < L) ? (p.mask != nullptr ? p.mask[row0 + i] : 0.f) : -INFINITY;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  const int l31 = lane & 31, hh = lane >> 5;
  const int nb = Lp >> 5;
  const LaneOffs lo_ = make_lane_offs(lane);
  for (int qb = wave; qb < nb; qb += 4) {
    bf16x8 qf[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) qf[ks] = row_frag_o(tQ, lo_, qb, ks);
    // pass 1: running max / sum of this lane's query column
    float m_run = -1e30f, l_run = 0.f;
    for (int kb = 0; kb < nb; ++kb) {
      f32x16 st = zero16();
#pragma unroll
      for (int ks = 0; ks < 4; ++ks)
        st = __builtin_amdgcn_mfma_f32_32x32x16_bf16(row_frag_o(tK, lo_, kb, ks), qf[ks], st, 0, 0, 0);
      float bm = -1e30f, mk[16];
      load_rows16(maskv, kb, hh, mk);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        st[r] = st[r] * 0.125f + mk[r];
        bm = fmaxf(bm, st[r]);
      }
      const float m_new = fmaxf(m_run, bm);
      float acc = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc += __expf(st[r] - m_new);
      l_run = l_run * __expf(m_run - m_new) + acc;
      m_run = m_new;
    }
    {
      const float m_o = __shfl_xor(m_run, 32), l_o = __shfl_xor(l_run, 32);
      const float mm = fmaxf(m_run, m_o);
      l_run = l_run * __expf(m_run - mm) + l_o * __expf(m_o - mm);
      m_run = mm;
    }
    const float lse = m_run + __logf(l_run);
    const int q = 32 * qb + l31;
    if (p.lse != nullptr && hh == 0 && q < L) p.lse[(int64_t)bh * LT + q] = lse;

    // pass 2: P^T = exp(S^T - lse) (dropout), O^T += V^T · P^T
    f32x16 o[2] = {zero16(), zero16()};
    for (int kb = 0; kb < nb; ++kb) {
      f32x16 st = zero16();
#pragma unroll
      for (int ks = 0; ks < 4; ++ks)
        st = __builtin_amdgcn_mfma_f32_32x32x16_bf16(row_frag_o(tK, lo_, kb, ks), qf[ks], st, 0, 0, 0);
      float mk[16];
      load_rows16(maskv, kb, hh, mk);
#pragma unroll
      for (int r = 0; r < 16; ++r) st[r] = __expf(st[r] * 0.125f + mk[r] - lse);
      if (p.drop.thresh16 != 0) drop_block16<lo32>(p.drop, ((uint64_t)bh * LT + q) * (uint64_t)LpT + 32 * kb + 4 * hh, st);
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const bf16x8 pb = pack8(st, s);
#pragma unroll
        for (int db = 0; db < 2; ++db)
          o[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_frag_o(tV, lo_, 2 * kb + s, db), pb, o[db], 0, 0, 0);
      }
    }
    // block qb of the Q tile is this wave's alone (its fragments went into registers above): the transpose buffer
    put_block32(tQ + qb * 4096, o, 1.0f, l31, hh);
    store_block32(tQ + qb * 4096, p.ctx + (row0 + 32 * qb) * H + hd * 64, H, L - 32 * qb, lane);
  }
}

// ----------------------------------------------------------------------------------- backward
template <bool LO32>
__global__ __launch_bounds__(256, 2) void attn_bwd_kernel(AttnArgs p) {
  p.drop = drop_resolve(p.drop);      // device-side salt of graph-replayed steps (common.h)
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int bh = blockIdx.x;
  const int b = bh / p.heads, hd = bh - b * p.heads;
  const int LT = p.L, LpT = p.Lp, H = p.heads * 64;
  constexpr bool lo32 = LO32;      // every dropout element index fits 32 bits (chosen by the launch: B heads L Lp < 2^32)
  const int L = (p.seq_len != nullptr) ? p.seq_len[b] : LT;
  if (L <= 0) return;
#ifdef MVPTR_TIMELINE_BUILD
  const unsigned long long tl0 = __builtin_amdgcn_s_memrealtime();
#endif
  const int64_t row0 = (p.seq_start != nullptr) ? (int64_t)p.seq_start[b] : (int64_t)b * LT;
  const int Lp = (L + 31) & ~31;
  const int64_t ldq = 3 * (int64_t)H;
  char* tQ = smem;
  char* tK = tQ + LpT * 128;
  char* tV = tK + LpT * 128;
  char* tD = tV + LpT * 128;  // dO
  float* maskv = reinterpret_cast<float*>(tD + LpT * 128);
  float* lsev = maskv + LpT;
  float* deltav = lsev + LpT;

  const __bf16* base = p.qkv + row0 * ldq + hd * 64;
  const __bf16* dob = p.dctx + row0 * H + hd * 64;
  const __bf16* ob = p.ctx + row0 * H + hd * 64;
  stage_tile(tQ, base, ldq, L, Lp, wave, lane);
  stage_tile(tK, base + H, ldq, L, Lp, wave, lane);
  stage_tile(tV, base + 2 * H, ldq, L, Lp, wave, lane);
  stage_tile(tD, dob, H, L, Lp, wave, lane);
  for (int i = tid; i < Lp; i += 256) {
    float mk = -INFINITY, ls = 0.f, dl = 0.f;
    if (i < L) {
      mk = (p.mask != nullptr) ? p.mask[row0 + i] : 0.f;
      ls = p.lse[(int64_t)bh * LT + i];
      const bf16x8* a = reinterpret_cast<const bf16x8*>(dob + (int64_t)i * H);
      const bf16x8* c = reinterpret_cast<const bf16x8*>(ob + (int64_t)i * H);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const bf16x8 x = a[j], y = c[j];
#pragma unroll
        for (int e = 0; e < 8; ++e) dl += bf2f(x[e]) * bf2f(y[e]);
      }
    }
    maskv[i] = mk;
    lsev[i] = ls;
    deltav[i] = dl;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  const int l31 = lane & 31, hh = lane >> 5;
  const int nb = Lp >> 5;
  const LaneOffs lo_ = make_lane_offs(lane);
  __bf16* dq_base = p.dqkv + row0 * ldq + hd * 64;
#ifdef MVPTR_TIMELINE_BUILD
  const unsigned long long tl1 = __builtin_amdgcn_s_memrealtime();
#endif

  // 2 nb independent tasks dealt to the four waves: task t < nb is "pass A" for key block t, task
  // t >= nb is "pass B" for query block t - nb.  One list instead of two loops keeps all four waves busy
  // on the short sequences of this path: with nb = 1 or 2 (L <= 64) the two passes take one round of the
  // waves instead of two, with nb = 5 three instead of four.
  for (int task = wave; task < 2 * nb; task += 4) {
  if (task < nb) {
    // ---------------- pass A: the wave owns a key block; key on lane, query in registers
    const int kb = task;
    const int key = 32 * kb + l31;
    const float mk = maskv[key];
    f32x16 dk[2] = {zero16(), zero16()}, dv[2] = {zero16(), zero16()};
    for (int qb = 0; qb < nb; ++qb) {
      f32x16 s = zero16(), dp = zero16();
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(row_frag_o(tQ, lo_, qb, ks), row_frag_o(tK, lo_, kb, ks), s, 0, 0, 0);
        dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(row_frag_o(tD, lo_, qb, ks), row_frag_o(tV, lo_, kb, ks), dp, 0, 0, 0);
      }
      // dropout keep bits: the pair (key & ~1, key | 1) of query q shares one hash; this lane
      // hashes the queries whose register parity equals its key parity and swaps with lane ^ 1
      uint32_t keepbits = 0xffffu;
      if (p.drop.thresh16 != 0) {
        keepbits = keep_bits16<lo32>(p.drop, bh, LT, LpT, qb, hh, key);
      }
#pragma unroll
      for (int t4 = 0; t4 < 4; ++t4) {
        const f32x4 ls4 = *reinterpret_cast<const f32x4*>(lsev + 32 * qb + 8 * t4 + 4 * hh);
        const f32x4 dl4 = *reinterpret_cast<const f32x4*>(deltav + 32 * qb + 8 * t4 + 4 * hh);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int r = 4 * t4 + e;
          const float pr = __expf(s[r] * 0.125f + mk - ls4[e]);
          float pd = pr, dpp = dp[r];
          if (p.drop.thresh16 != 0) {
            const bool keep = (keepbits >> r) & 1u;
            pd = keep ? pr * p.drop.scale : 0.f;
            dpp = keep ? dpp * p.drop.scale : 0.f;
          }
          s[r] = pd;                        // dropped-out probabilities (for dV)
          dp[r] = pr * (dpp - dl4[e]);      // dS
        }
      }
#pragma unroll
      for (int st = 0; st < 2; ++st) {
        const bf16x8 pb = pack8(s, st), dsb = pack8(dp, st);
        const int blk16 = 2 * qb + st;
#pragma unroll
        for (int db = 0; db < 2; ++db) {
          dv[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_frag_o(tD, lo_, blk16, db), pb, dv[db], 0, 0, 0);
          dk[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_frag_o(tQ, lo_, blk16, db), dsb, dk[db], 0, 0, 0);
        }
      }
    }
    if (key < L) {
      __bf16* dK = dq_base + (int64_t)key * ldq + H;
      __bf16* dV = dq_base + (int64_t)key * ldq + 2 * H;
#pragma unroll
      for (int db = 0; db < 2; ++db)
#pragma unroll
        for (int t4 = 0; t4 < 4; ++t4) {
          const int d = 32 * db + 8 * t4 + 4 * hh;
          bf16x4 a = {f2bf(dk[db][4 * t4] * 0.125f), f2bf(dk[db][4 * t4 + 1] * 0.125f),
                      f2bf(dk[db][4 * t4 + 2] * 0.125f), f2bf(dk[db][4 * t4 + 3] * 0.125f)};
          bf16x4 c = {f2bf(dv[db][4 * t4]), f2bf(dv[db][4 * t4 + 1]), f2bf(dv[db][4 * t4 + 2]), f2bf(dv[db][4 * t4 + 3])};
          *reinterpret_cast<bf16x4*>(dK + d) = a;
          *reinterpret_cast<bf16x4*>(dV + d) = c;
        }
    }
  } else {
    // ---------------- pass B: the wave owns a query block; query on lane, key in registers
    const int qb = task - nb;
    const int q = 32 * qb + l31;
    const float ls = lsev[q], dl = deltav[q];
    f32x16 dq[2] = {zero16(), zero16()};
    for (int kb = 0; kb < nb; ++kb) {
      f32x16 s = zero16(), dp = zero16();
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(row_frag_o(tK, lo_, kb, ks), row_frag_o(tQ, lo_, qb, ks), s, 0, 0, 0);
        dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(row_frag_o(tV, lo_, kb, ks), row_frag_o(tD, lo_, qb, ks), dp, 0, 0, 0);
      }
      float mk16[16];
      load_rows16(maskv, kb, hh, mk16);
      if (p.drop.thresh16 != 0) drop_block16<lo32>(p.drop, ((uint64_t)bh * LT + q) * (uint64_t)LpT + 32 * kb + 4 * hh, dp);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float pr = __expf(s[r] * 0.125f + mk16[r] - ls);
        s[r] = pr * (dp[r] - dl);  // dS^T
      }
#pragma unroll
      for (int st = 0; st < 2; ++st) {
        const bf16x8 dsb = pack8(s, st);
#pragma unroll
        for (int db = 0; db < 2; ++db)
          dq[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_frag_o(tK, lo_, 2 * kb + st, db), dsb, dq[db], 0, 0, 0);
      }
    }
    if (q < L) {
      __bf16* dQ = dq_base + (int64_t)q * ldq;
#pragma unroll
      for (int db = 0; db < 2; ++db)
#pragma unroll
        for (int t4 = 0; t4 < 4; ++t4) {
          bf16x4 a = {f2bf(dq[db][4 * t4] * 0.125f), f2bf(dq[db][4 * t4 + 1] * 0.125f),
                      f2bf(dq[db][4 * t4 + 2] * 0.125f), f2bf(dq[db][4 * t4 + 3] * 0.125f)};
          *reinterpret_cast<bf16x4*>(dQ + 32 * db + 8 * t4 + 4 * hh) = a;
        }
    }
  }
  }
#ifdef MVPTR_TIMELINE_BUILD
  {
    // diagnostic build: 100-MHz ticks at entry / after the staging barrier / when this wave is done (wave 0 and
    // the slowest wave by atomicMax) -> prologue and task-loop times per workgroup; a buffer of its own
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long tl2 = __builtin_amdgcn_s_memrealtime();
    if (p.stamps != nullptr && lane == 0) {
      unsigned long long* o = p.stamps + (size_t)blockIdx.x * 8;
      if (wave == 0) {
        o[0] = tl0;
        o[1] = tl1;
        o[3] = (unsigned long long)L;
      }
      atomicMax(o + 2, tl2);
    }
  }
#endif
}

// ------------------------------------------------------------- backward, sequences of <= 128 rows
// One pass instead of two: wave w owns key block w (key on lane) and computes P and dS once per (query block, key
// block) pair — dV^T += dO^T·P and dK^T += Q^T·dS stay in its registers, and its dS blocks (bf16, the B operand it
// has just fed to the dK product) are kept until every wave is done with V and dO; they then go into LDS over those
// two tiles as dS[key][query] in the tiles' own row format, and wave w, now owner of QUERY block w, reads them
// back transposed (ds_read_b64_tr_b16) as the B operand of dQ^T += K^T·dS^T.  Against the two-pass kernel above
// (which recomputes S, dP, the exponentials and the dropout hashes with the query on the lane to get dS^T in
// registers) a block pair costs 20 MFMAs, 16 exponentials and 8 hashes instead of 28 / 32 / 16.
// Needs all of a sequence's blocks resident at once: Lp <= 128 (four waves, one key block each).
// OCC workgroups per CU: 3 where the LDS footprint allows it (Lp <= 96: the text and visual stacks, whose workgroups
// are short and latency-bound: -15 % on their launches), 2 for Lp = 128 (the register budget of 3 costs spills there).
template <int OCC, bool LO32, int NBMAX = 4>
__global__ __launch_bounds__(256, OCC) void attn_bwd_fused_kernel(AttnArgs p) {
  p.drop = drop_resolve(p.drop);      // device-side salt of graph-replayed steps (common.h)
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int bh = blockIdx.x;
  const int b = bh / p.heads, hd = bh - b * p.heads;
  const int LT = p.L, LpT = p.Lp, H = p.heads * 64;
  constexpr bool lo32 = LO32;      // every dropout element index fits 32 bits (chosen by the launch: B heads L Lp < 2^32)
  const int L = (p.seq_len != nullptr) ? p.seq_len[b] : LT;
  if (L <= 0) return;
#ifdef MVPTR_TIMELINE_BUILD
  const unsigned long long tl0 = __builtin_amdgcn_s_memrealtime();
#endif
  const int64_t row0 = (p.seq_start != nullptr) ? (int64_t)p.seq_start[b] : (int64_t)b * LT;
  const int Lp = (L + 31) & ~31;
  const int64_t ldq = 3 * (int64_t)H;
  char* tQ = smem;
  char* tK = tQ + LpT * 128;
  char* tV = tK + LpT * 128;
  char* tD = tV + LpT * 128;  // dO
  float* maskv = reinterpret_cast<float*>(tD + LpT * 128);
  float* lsev = maskv + LpT;
  float* deltav = lsev + LpT;

  const __bf16* base = p.qkv + row0 * ldq + hd * 64;
  const __bf16* dob = p.dctx + row0 * H + hd * 64;
  const __bf16* ob = p.ctx + row0 * H + hd * 64;
  stage_tile(tQ, base, ldq, L, Lp, wave, lane);
  stage_tile(tK, base + H, ldq, L, Lp, wave, lane);
  stage_tile(tV, base + 2 * H, ldq, L, Lp, wave, lane);
  stage_tile(tD, dob, H, L, Lp, wave, lane);
  for (int i = tid; i < Lp; i += 256) {
    maskv[i] = (i < L) ? ((p.mask != nullptr) ? p.mask[row0 + i] : 0.f) : -INFINITY;
    lsev[i] = (i < L) ? p.lse[(int64_t)bh * LT + i] : 0.f;
  }
  // delta[q] = dO[q] . O[q]: 8 lanes per row, 16 bytes each (a wave instruction reads 8 whole lines; one lane per row
  // touched 64 lines per instruction: 24 us of the joint stack's 196, profiles/r04_experiments.txt)
  for (int g8 = wave; g8 < (Lp >> 3); g8 += 4) {
    const int row = 8 * g8 + (lane >> 3), c = lane & 7;
    float dl = 0.f;
    if (row < L) {
      const bf16x8 x = *reinterpret_cast<const bf16x8*>(dob + (int64_t)row * H + 8 * c);
      const bf16x8 y = *reinterpret_cast<const bf16x8*>(ob + (int64_t)row * H + 8 * c);
#pragma unroll
      for (int e = 0; e < 8; ++e) dl += bf2f(x[e]) * bf2f(y[e]);
    }
    dl = MVPTR_DPP_ADD(dl, 0xB1);    // quad_perm [1,0,3,2]
    dl = MVPTR_DPP_ADD(dl, 0x4E);    // quad_perm [2,3,0,1]
    dl = MVPTR_DPP_ADD(dl, 0x141);   // row_half_mirror: the other quad of the 8 lanes
    if (c == 0) deltav[row] = dl;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  const int l31 = lane & 31, hh = lane >> 5;
  const int nb = Lp >> 5;  // <= 4
  const LaneOffs lo_ = make_lane_offs(lane);
  __bf16* dq_base = p.dqkv + row0 * ldq + hd * 64;
#ifdef MVPTR_TIMELINE_BUILD
  const unsigned long long tl1 = __builtin_amdgcn_s_memrealtime();
#endif
  const bool owner = wave < nb;
  const int key = 32 * wave + l31;
  f32x16 dk[2] = {zero16(), zero16()}, dv[2] = {zero16(), zero16()};
  bf16x8 ds_keep[NBMAX][2];      // NBMAX: 32-row blocks a sequence of this launch can have (3 for Lp <= 96: 12 -> 7 spilled registers at three workgroups per CU)
  if (owner) {
    // ---------------- phase 1: key block `wave`; key on lane, queries in registers
    const int kb = wave;
    const float mk = maskv[key];
#pragma unroll
    for (int qb = 0; qb < NBMAX; ++qb) {
      if (qb < nb) {
        f32x16 s = zero16(), dp = zero16();
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
          s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(row_frag_o(tQ, lo_, qb, ks), row_frag_o(tK, lo_, kb, ks), s, 0, 0, 0);
          dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(row_frag_o(tD, lo_, qb, ks), row_frag_o(tV, lo_, kb, ks), dp, 0, 0, 0);
        }
        // dropout keep bits: see attn_bwd_kernel pass A
        uint32_t keepbits = 0xffffu;
        if (p.drop.thresh16 != 0) {
          if constexpr (OCC >= 3) {
            // three workgroups per CU leave 168 registers: the hashes one after the other, as rounds 3-5 had them (with the chains
            // in lockstep — or merely in one basic block — hipcc overlaps them and spills 26 registers: 169 -> 198 us at L = 96)
            keepbits = 0;
            const int par = key & 1;
#pragma unroll
            for (int r0 = 0; r0 < 16; r0 += 2) {
              const int rm = r0 + par;
              const int qm = 32 * qb + (rm & 3) + 8 * (rm >> 2) + 4 * hh;
              const uint64_t idx = ((uint64_t)bh * LT + qm) * (uint64_t)LpT + (key & ~1);
              const uint32_t hm = mvptr_pair_hash(idx >> 1, p.drop.seed_lo, p.drop.seed_hi);
              const uint32_t ho = (uint32_t)__builtin_amdgcn_mov_dpp((int)hm, 0xB1, 0xF, 0xF, true);  // quad_perm [1,0,3,2]
              const uint32_t um = par ? (hm >> 16) : (hm & 0xffffu);
              const uint32_t uo = par ? (ho >> 16) : (ho & 0xffffu);
              keepbits |= (um >= p.drop.thresh16 ? 1u : 0u) << rm;
              keepbits |= (uo >= p.drop.thresh16 ? 1u : 0u) << (rm ^ 1);
            }
          } else {
            keepbits = keep_bits16<lo32>(p.drop, bh, LT, LpT, qb, hh, key);
          }
        }
#pragma unroll
        for (int t4 = 0; t4 < 4; ++t4) {
          const f32x4 ls4 = *reinterpret_cast<const f32x4*>(lsev + 32 * qb + 8 * t4 + 4 * hh);
          const f32x4 dl4 = *reinterpret_cast<const f32x4*>(deltav + 32 * qb + 8 * t4 + 4 * hh);
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int r = 4 * t4 + e;
            const float pr = __expf(s[r] * 0.125f + mk - ls4[e]);
            float pd = pr, dpp = dp[r];
            if (p.drop.thresh16 != 0) {
              const bool keep = (keepbits >> r) & 1u;
              pd = keep ? pr * p.drop.scale : 0.f;
              dpp = keep ? dpp * p.drop.scale : 0.f;
            }
            s[r] = pd;                    // dropped-out probabilities (for dV)
            dp[r] = pr * (dpp - dl4[e]);  // dS
          }
        }
#pragma unroll
        for (int st = 0; st < 2; ++st) {
          const bf16x8 pb = pack8(s, st), dsb = pack8(dp, st);
          ds_keep[qb][st] = dsb;
          const int blk16 = 2 * qb + st;
#pragma unroll
          for (int db = 0; db < 2; ++db) {
            dv[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_frag_o(tD, lo_, blk16, db), pb, dv[db], 0, 0, 0);
            dk[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_frag_o(tQ, lo_, blk16, db), dsb, dk[db], 0, 0, 0);
          }
        }
      }
    }
  }
#ifdef MVPTR_TIMELINE_BUILD
  const unsigned long long tl_p1 = __builtin_amdgcn_s_memrealtime();
#endif
  __syncthreads();  // nobody reads V / dO any more
  if (owner) {
    // dS[key][query] over the V (queries 0..63) and dO (queries 64..127) tiles: row = key, 128-B rows, same chunk swizzle;
    // register e of pack8(., st) is query 32 qb + 16 st + 8 (e >> 2) + 4 hh + (e & 3)
#pragma unroll
    for (int qb = 0; qb < NBMAX; ++qb) {
      if (qb < nb) {
        char* ts = (qb >> 1) ? tD : tV;
#pragma unroll
        for (int st = 0; st < 2; ++st) {
          const bf16x4 lo4 = {ds_keep[qb][st][0], ds_keep[qb][st][1], ds_keep[qb][st][2], ds_keep[qb][st][3]};
          const bf16x4 hi4 = {ds_keep[qb][st][4], ds_keep[qb][st][5], ds_keep[qb][st][6], ds_keep[qb][st][7]};
          const int ch = 4 * (qb & 1) + 2 * st;
          *reinterpret_cast<bf16x4*>(ts + tile_off(key, ch) + 8 * hh) = lo4;
          *reinterpret_cast<bf16x4*>(ts + tile_off(key, ch + 1) + 8 * hh) = hi4;
        }
      }
    }
  }
  __syncthreads();
  if (owner) {
    // Q is not read after phase 1: block `wave` of its tile is this wave's transpose buffer for dK, dV and dQ
    char* tb = tQ + wave * 4096;
    __bf16* out0 = dq_base + (int64_t)(32 * wave) * ldq;
    put_block32(tb, dk, 0.125f, l31, hh);
    store_block32(tb, out0 + H, ldq, L - 32 * wave, lane);
    put_block32(tb, dv, 1.0f, l31, hh);
    store_block32(tb, out0 + 2 * H, ldq, L - 32 * wave, lane);
    // ---------------- phase 2: query block `wave`; query on lane: dQ^T += K^T · dS^T over all key blocks
    const int qb = wave;
    const char* ts = (qb >> 1) ? tD : tV;
    f32x16 dq[2] = {zero16(), zero16()};
    for (int kb = 0; kb < nb; ++kb) {
#pragma unroll
      for (int st = 0; st < 2; ++st) {
        const bf16x8 dsb = tr_frag_o(ts, lo_, 2 * kb + st, qb & 1);
#pragma unroll
        for (int db = 0; db < 2; ++db)
          dq[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_frag_o(tK, lo_, 2 * kb + st, db), dsb, dq[db], 0, 0, 0);
      }
    }
    put_block32(tb, dq, 0.125f, l31, hh);
    store_block32(tb, out0, ldq, L - 32 * wave, lane);
  }
#ifdef MVPTR_TIMELINE_BUILD
  {
    // 100-MHz ticks: entry / staging barrier passed / this wave's phase 1 done (slowest by atomicMax) / wave done
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long tl2 = __builtin_amdgcn_s_memrealtime();
    if (p.stamps != nullptr && lane == 0) {
      unsigned long long* o = p.stamps + (size_t)blockIdx.x * 8;
      if (wave == 0) {
        o[0] = tl0;
        o[1] = tl1;
        o[3] = (unsigned long long)L;
      }
      atomicMax(o + 2, tl2);
      atomicMax(o + 4, tl_p1);
    }
  }
#endif
}

// ------------------------------------------------------- the probabilities as a tensor (output_attentions)
// modeling_vlbert.py:100 hands attention_probs back when config.output_attentions is set: an inspection path, not part of
// a step (the step's kernels above never write anything of size L x L).  One wave per (batch, head, query): a lane owns the
// keys lane, lane + 64, ...; scores in f32 from the bf16 Q / K rows, softmax over the row, f32 out [B, heads, L, L].
__global__ __launch_bounds__(256) void attn_probs_kernel(const __bf16* __restrict__ qkv, const float* __restrict__ mask,
                                                          float* __restrict__ probs, int B, int L, int heads) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);      // (b * heads + h) * L + q
  if (row >= (int64_t)B * heads * L) return;
  const int q = (int)(row % L);
  const int bh = (int)(row / L);
  const int h = bh % heads, b = bh / heads;
  const int64_t ldq = 3 * (int64_t)heads * 64;
  const __bf16* qp = qkv + ((int64_t)b * L + q) * ldq + h * 64;
  float qv[64];
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    const bf16x8 x = *reinterpret_cast<const bf16x8*>(qp + 8 * c);
#pragma unroll
    for (int e = 0; e < 8; ++e) qv[8 * c + e] = bf2f(x[e]);
  }
  float s[4];
  float mx = -INFINITY;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int key = lane + 64 * j;
    s[j] = -INFINITY;
    if (key < L) {
      const __bf16* kp = qkv + ((int64_t)b * L + key) * ldq + heads * 64 + h * 64;
      float d = 0.f;
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        const bf16x8 x = *reinterpret_cast<const bf16x8*>(kp + 8 * c);
#pragma unroll
        for (int e = 0; e < 8; ++e) d += qv[8 * c + e] * bf2f(x[e]);
      }
      s[j] = d * 0.125f + mask[(int64_t)b * L + key];
    }
    mx = fmaxf(mx, s[j]);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
  float sum = 0.f;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    s[j] = (lane + 64 * j < L) ? __expf(s[j] - mx) : 0.f;
    sum += s[j];
  }
  sum = wave_sum(sum);
  const float inv = 1.f / sum;
#pragma unroll
  for (int j = 0; j < 4; ++j)
    if (lane + 64 * j < L) probs[row * L + lane + 64 * j] = s[j] * inv;
}

// diagnostic build only: MVPTR_ATTN_TWO_PASS=1 keeps the two-pass kernel for every length (A/B measurements)
bool two_pass_forced() {
#ifdef MVPTR_DIAG_BUILD
  static const bool v = [] {
    const char* e = getenv("MVPTR_ATTN_TWO_PASS");
    return e && e[0] == '1';
  }();
  return v;
#else
  return false;
#endif
}

int check_common(const char* who, const void* qkv, int B, int L, int heads) {
  if (B <= 0 || L <= 0 || heads <= 0) MVPTR_FAIL(MVPTR_BAD_SHAPE, "%s: B, L, heads must be > 0", who);
  if (L > 256) MVPTR_FAIL(MVPTR_BAD_SHAPE, "%s: L=%d > 256 not supported", who, L);
  if ((uintptr_t)qkv & 15) MVPTR_FAIL(MVPTR_BAD_ALIGN, "%s: qkv must be 16-byte aligned", who);
  return MVPTR_OK;
}

}  // namespace

extern "C" int mvptr_attention_fwd_packed(const void* qkv, const float* mask_add, void* ctx, float* lse,
                                          const int* seq_start, const int* seq_len, int B, int L,
                                          int heads, const mvptr_dropout* drop, void* stream) {
  int rc = check_common("attention_fwd", qkv, B, L, heads);
  if (rc) return rc;
  if (!ctx) MVPTR_FAIL(MVPTR_BAD_ARG, "attention_fwd: NULL ctx");
  if ((uintptr_t)ctx & 15) MVPTR_FAIL(MVPTR_BAD_ALIGN, "attention_fwd: ctx must be 16-byte aligned (rows leave as 16-byte pieces)");
  if ((seq_start == nullptr) != (seq_len == nullptr)) MVPTR_FAIL(MVPTR_BAD_ARG, "attention_fwd: seq_start and seq_len go together");
  if (!mask_add && !seq_len) MVPTR_FAIL(MVPTR_BAD_ARG, "attention_fwd: NULL mask (only allowed in packed mode)");
  AttnArgs a{};
  a.qkv = (const __bf16*)qkv;
  a.mask = mask_add;
  a.ctx = (__bf16*)ctx;
  a.lse = lse;
  a.B = B;
  a.L = L;
  a.heads = heads;
  a.Lp = (L + 31) & ~31;
  a.seq_start = seq_start;
  a.seq_len = seq_len;
  a.drop = make_dropdev(drop);
  const size_t lds = (size_t)a.Lp * 128 * 3 + (size_t)a.Lp * 4;
  const bool lo32 = (uint64_t)B * (uint64_t)heads * (uint64_t)a.L * (uint64_t)a.Lp < ((uint64_t)1 << 32);   // dropout element indices in 32 bits
  hipError_t e = hipFuncSetAttribute(lo32 ? (const void*)attn_fwd_kernel<true> : (const void*)attn_fwd_kernel<false>,
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) MVPTR_FAIL(MVPTR_HIP_ERROR, "attention_fwd: set LDS size: %s", hipGetErrorString(e));
  if (lo32) hipLaunchKernelGGL(attn_fwd_kernel<true>, dim3(B * heads), dim3(256), lds, (hipStream_t)stream, a);
  else hipLaunchKernelGGL(attn_fwd_kernel<false>, dim3(B * heads), dim3(256), lds, (hipStream_t)stream, a);
  MVPTR_CHECK_LAUNCH("attention_fwd");
  return MVPTR_OK;
}

extern "C" int mvptr_attention_fwd(const void* qkv, const float* mask_add, void* ctx, float* lse,
                                   int B, int L, int heads, const mvptr_dropout* drop,
                                   void* stream) {
  if (!mask_add) MVPTR_FAIL(MVPTR_BAD_ARG, "attention_fwd: NULL mask/ctx");
  return mvptr_attention_fwd_packed(qkv, mask_add, ctx, lse, nullptr, nullptr, B, L, heads, drop, stream);
}

extern "C" int mvptr_attention_probs(const void* qkv, const float* mask_add, float* probs, int B, int L, int heads, void* stream) {
  int rc = check_common("attention_probs", qkv, B, L, heads);
  if (rc) return rc;
  if (!mask_add || !probs) MVPTR_FAIL(MVPTR_BAD_ARG, "attention_probs: NULL argument");
  const int64_t rows = (int64_t)B * heads * L;
  if ((rows + 3) / 4 > 0x7fffffff) MVPTR_FAIL(MVPTR_BAD_SHAPE, "attention_probs: too many rows");
  hipLaunchKernelGGL(attn_probs_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, (const __bf16*)qkv,
                     mask_add, probs, B, L, heads);
  MVPTR_CHECK_LAUNCH("attention_probs");
  return MVPTR_OK;
}

extern "C" int mvptr_attention_bwd_packed(const void* qkv, const float* mask_add, const void* ctx,
                                          const void* dctx, const float* lse, void* dqkv,
                                          const int* seq_start, const int* seq_len, int B, int L,
                                          int heads, const mvptr_dropout* drop, void* stream) {
  int rc = check_common("attention_bwd", qkv, B, L, heads);
  if (rc) return rc;
  if (!ctx || !dctx || !lse || !dqkv) MVPTR_FAIL(MVPTR_BAD_ARG, "attention_bwd: NULL argument");
  if ((seq_start == nullptr) != (seq_len == nullptr)) MVPTR_FAIL(MVPTR_BAD_ARG, "attention_bwd: seq_start and seq_len go together");
  if (!mask_add && !seq_len) MVPTR_FAIL(MVPTR_BAD_ARG, "attention_bwd: NULL mask (only allowed in packed mode)");
  if (((uintptr_t)dctx & 15) || ((uintptr_t)ctx & 15) || ((uintptr_t)dqkv & 15))
    MVPTR_FAIL(MVPTR_BAD_ALIGN, "attention_bwd: ctx / dctx / dqkv must be 16-byte aligned");
  AttnArgs a{};
  a.qkv = (const __bf16*)qkv;
  a.mask = mask_add;
  a.ctx = (__bf16*)const_cast<void*>(ctx);
  a.dctx = (const __bf16*)dctx;
  a.lse = const_cast<float*>(lse);
  a.dqkv = (__bf16*)dqkv;
  a.B = B;
  a.L = L;
  a.heads = heads;
  a.Lp = (L + 31) & ~31;
  a.seq_start = seq_start;
  a.seq_len = seq_len;
  a.drop = make_dropdev(drop);
  a.stamps = nullptr;
#ifdef MVPTR_TIMELINE_BUILD
  a.stamps = (unsigned long long*)mvptr_knobs().stamps;
#endif
  const size_t lds = (size_t)a.Lp * 128 * 4 + (size_t)a.Lp * 12;
  // sequences of <= 128 rows (every stack of the pre-training step): the one-pass kernel; longer ones: two passes
  const bool fused = a.Lp <= 128 && !two_pass_forced();
  const bool occ3 = fused && a.Lp <= 96;
  const bool lo32 = (uint64_t)B * (uint64_t)heads * (uint64_t)a.L * (uint64_t)a.Lp < ((uint64_t)1 << 32);   // dropout element indices in 32 bits
  auto launch = [&](auto kernel) -> int {
    hipError_t e = hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) MVPTR_FAIL(MVPTR_HIP_ERROR, "attention_bwd: set LDS size: %s", hipGetErrorString(e));
    hipLaunchKernelGGL(kernel, dim3(B * heads), dim3(256), lds, (hipStream_t)stream, a);
    return MVPTR_OK;
  };
  int lrc;
  if (!fused) lrc = lo32 ? launch(attn_bwd_kernel<true>) : launch(attn_bwd_kernel<false>);
  else if (occ3) lrc = lo32 ? launch(attn_bwd_fused_kernel<3, true, 3>) : launch(attn_bwd_fused_kernel<3, false, 3>);
  else lrc = lo32 ? launch(attn_bwd_fused_kernel<2, true>) : launch(attn_bwd_fused_kernel<2, false>);
  if (lrc != MVPTR_OK) return lrc;
  MVPTR_CHECK_LAUNCH("attention_bwd");
  return MVPTR_OK;
}

extern "C" int mvptr_attention_bwd(const void* qkv, const float* mask_add, const void* ctx,
                                   const void* dctx, const float* lse, void* dqkv, int B, int L,
                                   int heads, const mvptr_dropout* drop, void* stream) {
  if (!mask_add) MVPTR_FAIL(MVPTR_BAD_ARG, "attention_bwd: NULL argument");
  return mvptr_attention_bwd_packed(qkv, mask_add, ctx, dctx, lse, dqkv, nullptr, nullptr, B, L, heads, drop, stream);
}
