// rows.hip — moving token rows between the encoder stacks and into the heads without materialising padded
// tensors: index maps built on the device from the attention masks, a row gather and its scatter-add.
//
// Replaces, for the row-packed execution of the stacks, the torch.cat / index_select / masked_select
// chains of oscar/modeling/modeling_vlbert.py:519 (only_vis slice), :544-552 (text + hard image, hard text + image),
// :586-590 (joint sequence), :1231-1234 and :1245 (masked rows for the MLM heads), transformers/
// pytorch_transformers/modeling_bert.py:471 ([CLS] row for the pooler) and the zero-fill + index_add kernels
// autograd derives for their backward passes.  HBM-bound byte moves: 16-byte pieces, one pass.
#include "common.h"

namespace {

// Sequence s of the output is the concatenation of up to two segments; segment k takes the slots
// [col0, col0 + len) of row sel[s] (s when sel == NULL) of its additive mask (valid slot <=> mask == 0) and
// names the source row of a valid slot: pos[sel * ld_pos + col] (a row of an already packed buffer) or, when
// pos == NULL, sel * src_seq_stride + col (a row of a padded [*, src_seq_stride, H] buffer), plus src_base.
// Three small launches: (1) one wave per sequence counts its valid slots (64 slots per ballot); (2) one
// workgroup scans the counts into seq_start and writes the totals; (3) one wave per sequence writes
// pos_out[s, slot] = packed row (-1: padded slot) and idx_out[packed row] = source row from ballot prefix counts.
__device__ __forceinline__ const float* seg_mask_row(const mvptr_pack_seg& g, int s, int64_t& r) {
  r = g.sel ? g.sel[s] : (int64_t)s;
  return g.mask + r * g.ld_mask + g.col0;
}

__global__ __launch_bounds__(256) void pack_count_kernel(mvptr_pack_seg s0, mvptr_pack_seg s1, int nseg, int n_seq,
                                                          int32_t* seq_len) {
  const int lane = threadIdx.x & 63;
  const int s = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (s >= n_seq) return;
  int cnt = 0;
  for (int k = 0; k < nseg; ++k) {
    const mvptr_pack_seg& g = k ? s1 : s0;
    int64_t r;
    const float* m = seg_mask_row(g, s, r);
    for (int l0 = 0; l0 < g.len; l0 += 64) {
      const bool ok = (l0 + lane < g.len) && (m[l0 + lane] == 0.f);
      cnt += __popcll(__ballot(ok));
    }
  }
  if (lane == 0) seq_len[s] = cnt;
}

__global__ __launch_bounds__(1024) void pack_scan_kernel(const int32_t* seq_len, int n_seq, int32_t* seq_start, int64_t* counts) {
  __shared__ int scan[1024];
  __shared__ int carry_s, maxlen_s;
  const int tid = threadIdx.x;
  if (tid == 0) {
    carry_s = 0;
    maxlen_s = 0;
  }
  __syncthreads();
  for (int base = 0; base < n_seq; base += 1024) {
    const int s = base + tid;
    const int cnt = (s < n_seq) ? seq_len[s] : 0;
    scan[tid] = cnt;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) {  // inclusive Hillis-Steele scan
      const int v = (tid >= o) ? scan[tid - o] : 0;
      __syncthreads();
      scan[tid] += v;
      __syncthreads();
    }
    const int carry = carry_s;
    if (s < n_seq) {
      seq_start[s] = carry + scan[tid] - cnt;
      atomicMax(&maxlen_s, cnt);
    }
    __syncthreads();
    if (tid == 1023) carry_s = carry + scan[1023];
    __syncthreads();
  }
  if (tid == 0) {
    counts[0] = carry_s;
    counts[1] = maxlen_s;
  }
}

__global__ __launch_bounds__(256) void pack_fill_kernel(mvptr_pack_seg s0, mvptr_pack_seg s1, int nseg, int n_seq,
                                                         const int32_t* seq_start, int32_t* pos_out, int32_t* idx_out) {
  const int lane = threadIdx.x & 63;
  const int s = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (s >= n_seq) return;
  const int Ltot = s0.len + (nseg > 1 ? s1.len : 0);
  int32_t* po = pos_out + (int64_t)s * Ltot;
  int run = seq_start[s];
  int slot0 = 0;
  const uint64_t lt = (lane == 0) ? 0ull : (~0ull >> (64 - lane));   // lanes below this one
  for (int k = 0; k < nseg; ++k) {
    const mvptr_pack_seg& g = k ? s1 : s0;
    int64_t r;
    const float* m = seg_mask_row(g, s, r);
    for (int l0 = 0; l0 < g.len; l0 += 64) {
      const int l = l0 + lane;
      const bool in = l < g.len;
      const bool ok = in && (m[l] == 0.f);
      const uint64_t bal = __ballot(ok);
      const int dest = run + __popcll(bal & lt);
      if (in) po[slot0 + l] = ok ? dest : -1;
      if (ok) {
        const int64_t src = g.pos ? (int64_t)g.pos[r * g.ld_pos + g.col0 + l] : r * g.src_seq_stride + g.col0 + l;
        idx_out[dest] = (int32_t)(src + g.src_base);
      }
      run += __popcll(bal);
    }
    slot0 += g.len;
  }
}

// out[i, :] = src[idx[i], :] (bf16 rows of H elements, 16-byte pieces); idx < 0: zero row; idx >= split reads
// row idx - split of src2 (two packed buffers addressed as one)
__global__ __launch_bounds__(256) void gather_rows_kernel(const __bf16* src, int64_t ld_src, const __bf16* src2, int64_t ld_src2,
                                                           int split, const int32_t* idx, __bf16* out, int64_t ld_out, int n,
                                                           int chunks) {
  const int64_t total = (int64_t)n * chunks;
  for (int64_t e = blockIdx.x * (int64_t)256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
    const int r = (int)(e / chunks), c = (int)(e - (int64_t)r * chunks);
    const int s = idx[r];
    u32x4 v = {0u, 0u, 0u, 0u};
    if (s >= 0) {
      const __bf16* sp = (src2 != nullptr && s >= split) ? src2 + (int64_t)(s - split) * ld_src2 : src + (int64_t)s * ld_src;
      v = *reinterpret_cast<const u32x4*>(sp + c * 8);
    }
    *reinterpret_cast<u32x4*>(out + (int64_t)r * ld_out + c * 8) = v;
  }
}
// dst[idx[i], :] += src[i, :]: rows may repeat and several calls may hit one row, so the adds are atomics — f32
// atomics into an f32 destination (dst_f32: sums exact to 2^-24 whatever the arrival order; the caller rounds to
// bf16 once) or packed-pair bf16 atomics into a bf16 destination (every add rounds: order-dependent at 2^-9).
// src is bf16, or f32 (gradient of an f32 head) when src_f32.
__global__ __launch_bounds__(256) void scatter_add_rows_kernel(const void* src, int64_t ld_src, int src_f32, const int32_t* idx,
                                                                void* dst_, int64_t ld_dst, void* dst2_, int64_t ld_dst2, int split,
                                                                int n, int pairs, int dst_f32) {
  const int64_t total = (int64_t)n * pairs;
  for (int64_t e = blockIdx.x * (int64_t)256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
    const int r = (int)(e / pairs), c = (int)(e - (int64_t)r * pairs);
    const int d = idx[r];
    if (d < 0) continue;
    float f0, f1;
    if (src_f32) {
      const float* sp = (const float*)src + (int64_t)r * ld_src + c * 2;
      f0 = sp[0];
      f1 = sp[1];
    } else {
      const bf16x2 sv = *reinterpret_cast<const bf16x2*>((const __bf16*)src + (int64_t)r * ld_src + c * 2);
      f0 = bf2f(sv[0]);
      f1 = bf2f(sv[1]);
    }
    const bool second = (dst2_ != nullptr && d >= split);
    const int64_t off = second ? (int64_t)(d - split) * ld_dst2 + c * 2 : (int64_t)d * ld_dst + c * 2;
    if (dst_f32) {
      float* dp = (float*)(second ? dst2_ : dst_) + off;
      atomicAdd(dp, f0);
      atomicAdd(dp + 1, f1);
    } else {
      __bf16* dp = (__bf16*)(second ? dst2_ : dst_) + off;
      bf16x2 v;
      v[0] = f2bf(f0);
      v[1] = f2bf(f1);
      asm volatile("global_atomic_pk_add_bf16 %0, %1, off" : : "v"(dp), "v"(v) : "memory");
    }
  }
}

// ---- gradient of the row taps in one pass over the tapped rows (round 4) ------------------------------------------
// The backward of a multi-tap gather used to be: zero an f32 [rows, H] buffer, one scatter-add launch per tap (f32
// atomics), cast the buffer to bf16 — 0.9 ms of a 26-ms step in 29 launches, most of it spent on rows nobody tapped
// (profiles/r04_experiments.txt).  Here the taps are inverted first — one launch threads every entry onto a linked
// list of its destination row (head + count per row, next per entry) — and one wave per destination row then sums ITS
// contributions in f32 and writes the bf16 row once (zeros for a row nobody tapped): no f32 buffer, no zero fill of
// the rows, no cast pass, no atomics on the rows.  A row with one contribution (the common case) costs two dependent
// loads (head, then the tapped row).  The contributions of a row are summed in ascending (tap, position) order whenever
// there are at most 64 of them (every case of the pre-training step), so the f32 sums — and the result — do not depend
// on the order in which the atomics of the list build arrived.
struct TapSet {
  mvptr_tap t[MVPTR_TAP_MAX];
  int base[MVPTR_TAP_MAX + 1];   // entries of tap k are [base[k], base[k+1])
  int count;
};
__device__ __forceinline__ int tap_of_entry(const TapSet& ts, int e) {
  int k = 0;
#pragma unroll
  for (int i = 1; i < MVPTR_TAP_MAX; ++i)
    if (i < ts.count && e >= ts.base[i]) k = i;
  return k;
}
// hc[r] = {entry + 1 at the head of row r's list (0: empty), number of entries}; next[e] = entry + 1 behind e
__global__ __launch_bounds__(256) void tap_link_kernel(TapSet ts, int R, int32_t* hc, int32_t* next) {
  const int total = ts.base[ts.count];
  for (int e = blockIdx.x * 256 + threadIdx.x; e < total; e += gridDim.x * 256) {
    const int k = tap_of_entry(ts, e);
    const int r = ts.t[k].idx[e - ts.base[k]];
    if (r >= 0 && r < R) {
      next[e] = atomicExch(hc + 2 * r, e + 1);
      atomicAdd(hc + 2 * r + 1, 1);
    }
  }
}
// one wave per destination row; lane l owns columns 4 l + 256 i
__global__ __launch_bounds__(256) void tap_sum_kernel(TapSet ts, int R, int split, const int32_t* hc, const int32_t* next,
                                                       __bf16* dst, int64_t ld_dst, __bf16* dst2, int64_t ld_dst2, int H) {
  const int lane = threadIdx.x & 63;
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= R) return;
  const int2 hn = *reinterpret_cast<const int2*>(hc + 2 * r);
  const int n = hn.y;
  __bf16* out = (dst2 != nullptr && r >= split) ? dst2 + (int64_t)(r - split) * ld_dst2 : dst + (int64_t)r * ld_dst;
  constexpr int MAXI = 8;                       // H <= 2048
  f32x4 acc[MAXI];
#pragma unroll
  for (int i = 0; i < MAXI; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  auto add_entry = [&](int e_v) {
    const int e = __builtin_amdgcn_readfirstlane(e_v);          // the same in every lane: scalar loads of the tap descriptor
    const int k = tap_of_entry(ts, e);
    const mvptr_tap& t = ts.t[k];
    const int64_t row = (int64_t)(e - ts.base[k]) * t.ld_g;
#pragma unroll
    for (int i = 0; i < MAXI; ++i) {
      const int c = 4 * lane + 256 * i;
      if (c < H) {
        if (t.g_f32) {
          const f32x4 v = *reinterpret_cast<const f32x4*>((const float*)t.g + row + c);
          acc[i] += v;
        } else {
          const bf16x4 v = *reinterpret_cast<const bf16x4*>((const __bf16*)t.g + row + c);
          acc[i] += f32x4{bf2f(v[0]), bf2f(v[1]), bf2f(v[2]), bf2f(v[3])};
        }
      }
    }
  };
  if (n == 1) {
    add_entry(hn.x - 1);
  } else if (n > 1 && n <= 64) {
    // walk the list into the lanes (entry q in lane q), then add in ascending entry order = (tap, position) order
    int mine = 0x7fffffff, e1 = hn.x;
    for (int q = 0; q < n; ++q) {
      if (lane == q) mine = e1 - 1;
      if (q + 1 < n) e1 = next[e1 - 1];
    }
    int rank = 0;
    for (int q = 0; q < n; ++q) rank += (__shfl(mine, q) < mine) ? 1 : 0;
    for (int q = 0; q < n; ++q) {
      const unsigned long long hit = __ballot(lane < n && rank == q);
      add_entry(__shfl(mine, (int)__ffsll((long long)hit) - 1));
    }
  } else if (n > 64) {
    int e1 = hn.x;
    for (int q = 0; q < n; ++q) {                 // more than 64 contributions: list (arrival) order
      const int e = e1 - 1;
      if (q + 1 < n) e1 = next[e];
      add_entry(e);
    }
  }
#pragma unroll
  for (int i = 0; i < MAXI; ++i) {
    const int c = 4 * lane + 256 * i;
    if (c < H) {
      const bf16x4 o = {f2bf(acc[i][0]), f2bf(acc[i][1]), f2bf(acc[i][2]), f2bf(acc[i][3])};
      *reinterpret_cast<bf16x4*>(out + c) = o;
    }
  }
}

}  // namespace

extern "C" int mvptr_pack_maps(const mvptr_pack_seg* segs, int nseg, int n_seq, int32_t* pos_out, int32_t* idx_out,
                               int32_t* seq_start, int32_t* seq_len, int64_t* counts, void* stream) {
  if (!segs || nseg < 1 || nseg > 2 || n_seq <= 0) MVPTR_FAIL(MVPTR_BAD_ARG, "pack_maps: 1 or 2 segments, n_seq > 0");
  if (!pos_out || !idx_out || !seq_start || !seq_len || !counts) MVPTR_FAIL(MVPTR_BAD_ARG, "pack_maps: NULL output");
  for (int k = 0; k < nseg; ++k)
    if (!segs[k].mask || segs[k].len <= 0 || segs[k].col0 < 0) MVPTR_FAIL(MVPTR_BAD_ARG, "pack_maps: bad segment %d", k);
  mvptr_pack_seg s1 = segs[nseg > 1 ? 1 : 0];
  const dim3 grid((n_seq + 3) / 4);
  hipLaunchKernelGGL(pack_count_kernel, grid, dim3(256), 0, (hipStream_t)stream, segs[0], s1, nseg, n_seq, seq_len);
  hipLaunchKernelGGL(pack_scan_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, seq_len, n_seq, seq_start, counts);
  hipLaunchKernelGGL(pack_fill_kernel, grid, dim3(256), 0, (hipStream_t)stream, segs[0], s1, nseg, n_seq, seq_start, pos_out, idx_out);
  MVPTR_CHECK_LAUNCH("pack_maps");
  return MVPTR_OK;
}

extern "C" int mvptr_gather_rows(const void* src, int64_t ld_src, const void* src2, int64_t ld_src2, int split,
                                 const int32_t* idx, void* out, int64_t ld_out, int n, int H, void* stream) {
  if (n <= 0 || H <= 0 || (H & 7) || (ld_src & 7) || (ld_out & 7) || (src2 && (ld_src2 & 7)))
    MVPTR_FAIL(MVPTR_BAD_SHAPE, "gather_rows: H and the leading dimensions must be positive multiples of 8");
  if (!src || !idx || !out || ((uintptr_t)src & 15) || ((uintptr_t)out & 15) || ((uintptr_t)src2 & 15))
    MVPTR_FAIL(MVPTR_BAD_ALIGN, "gather_rows: NULL or unaligned pointer");
  const int64_t total = (int64_t)n * (H / 8);
  int grid = (int)((total + 255) / 256);
  if (grid > 16384) grid = 16384;
  hipLaunchKernelGGL(gather_rows_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const __bf16*)src, ld_src,
                     (const __bf16*)src2, ld_src2, split, idx, (__bf16*)out, ld_out, n, H / 8);
  MVPTR_CHECK_LAUNCH("gather_rows");
  return MVPTR_OK;
}

extern "C" int mvptr_scatter_add_rows(const void* src, int64_t ld_src, int src_f32, const int32_t* idx, void* dst,
                                      int64_t ld_dst, void* dst2, int64_t ld_dst2, int split, int dst_f32, int n, int H,
                                      void* stream) {
  if (n <= 0 || H <= 0 || (H & 1) || (ld_src & 1) || (ld_dst & 1) || (dst2 && (ld_dst2 & 1)))
    MVPTR_FAIL(MVPTR_BAD_SHAPE, "scatter_add_rows: H and the leading dimensions must be positive and even");
  if (!src || !idx || !dst || ((uintptr_t)src & 3) || ((uintptr_t)dst & 3) || ((uintptr_t)dst2 & 3))
    MVPTR_FAIL(MVPTR_BAD_ALIGN, "scatter_add_rows: NULL or unaligned pointer");
  const int64_t total = (int64_t)n * (H / 2);
  int grid = (int)((total + 255) / 256);
  if (grid > 16384) grid = 16384;
  hipLaunchKernelGGL(scatter_add_rows_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, src, ld_src, src_f32 ? 1 : 0, idx,
                     dst, ld_dst, dst2, ld_dst2, split, n, H / 2, dst_f32 ? 1 : 0);
  MVPTR_CHECK_LAUNCH("scatter_add_rows");
  return MVPTR_OK;
}

extern "C" int mvptr_tap_rows_bwd(const mvptr_tap* taps, int ntaps, void* dst, int64_t ld_dst, int rows, void* dst2, int64_t ld_dst2,
                                  int rows2, int H, int32_t* work, int64_t work_elems, void* stream) {
  if (!taps || ntaps < 1 || ntaps > MVPTR_TAP_MAX) MVPTR_FAIL(MVPTR_BAD_ARG, "tap_rows_bwd: 1..%d taps", MVPTR_TAP_MAX);
  if (rows <= 0 || rows2 < 0 || H <= 0 || (H & 3) || H > 2048 || (ld_dst & 3) || (dst2 && (ld_dst2 & 3)))
    MVPTR_FAIL(MVPTR_BAD_SHAPE, "tap_rows_bwd: rows > 0, H a multiple of 4 and <= 2048, leading dimensions multiples of 4");
  if (!dst || !work || ((uintptr_t)dst & 7) || ((uintptr_t)dst2 & 7) || ((dst2 == nullptr) != (rows2 == 0)))
    MVPTR_FAIL(MVPTR_BAD_ALIGN, "tap_rows_bwd: NULL / unaligned destination, or dst2 without rows2");
  TapSet ts = {};
  ts.count = ntaps;
  int64_t total = 0;
  for (int k = 0; k < ntaps; ++k) {
    const mvptr_tap& t = taps[k];
    if (!t.g || !t.idx || t.n < 0 || t.n >= (1 << 24) || t.ld_g < H || (t.ld_g & 3) || ((uintptr_t)t.g & (t.g_f32 ? 15 : 7)))
      MVPTR_FAIL(MVPTR_BAD_ARG, "tap_rows_bwd: tap %d: NULL / unaligned rows, n outside [0, 2^24), or ld_g < H", k);
    ts.t[k] = t;
    ts.base[k] = (int)total;
    total += t.n;
  }
  if (total >= (int64_t)0x7fffffff) MVPTR_FAIL(MVPTR_BAD_SHAPE, "tap_rows_bwd: too many entries");
  ts.base[ntaps] = (int)total;
  const int R = rows + rows2;
  if (work_elems < 2 * (int64_t)(R + 1) + total) MVPTR_FAIL(MVPTR_BAD_SHAPE, "tap_rows_bwd: work needs 2 (rows + rows2 + 1) + entries int32");
  if ((uintptr_t)work & 7) MVPTR_FAIL(MVPTR_BAD_ALIGN, "tap_rows_bwd: work must be 8-byte aligned");
  int32_t* hc = work;                       // {head, count} per destination row
  int32_t* next = work + 2 * (R + 1);
  hipStream_t s = (hipStream_t)stream;
  if (hipMemsetAsync(hc, 0, sizeof(int32_t) * 2 * (size_t)R, s) != hipSuccess) MVPTR_FAIL(MVPTR_HIP_ERROR, "tap_rows_bwd: memset failed");
  if (total > 0) {
    const int g = (total + 255) / 256 > 4096 ? 4096 : (int)((total + 255) / 256);
    hipLaunchKernelGGL(tap_link_kernel, dim3(g), dim3(256), 0, s, ts, R, hc, next);
  }
  hipLaunchKernelGGL(tap_sum_kernel, dim3((R + 3) / 4), dim3(256), 0, s, ts, R, rows, hc, next, (__bf16*)dst, ld_dst, (__bf16*)dst2,
                     ld_dst2, H);
  MVPTR_CHECK_LAUNCH("tap_rows_bwd");
  return MVPTR_OK;
}


// Scored rows of a masked-LM head in one launch: the positions e of labels[B, L] (row-major) with label > -1, in ascending
// order, as out_labels[k] = labels[e] and out_rows[k] = pos[b * ld_pos + l] (the packed row of slot (b, l); pos NULL: e
// itself).  Exactly n_out entries are written: a surplus is cut, a shortfall padded with label -1 / row -1 (a zero row the
// loss ignores).  Replaces `keep = labels > -1; idx = nonzero_static(keep, size); labels.index_select(idx);
// pos.reshape(-1).index_select(idx)` — 7 small launches per head (oscar/modeling/modeling_vlbert.py:1231-1234,1245).
namespace {
__global__ __launch_bounds__(1024) void compact_scored_kernel(const int64_t* labels, const int32_t* pos, int64_t ld_pos, int B, int L,
                                                               int n_out, int64_t* out_labels, int32_t* out_rows,
                                                               unsigned long long* err) {
  __shared__ int scan[1024];
  const int tid = threadIdx.x, total = B * L;
  const int per = (total + 1023) / 1024;
  const int e0 = min(tid * per, total), e1 = min(e0 + per, total);
  // up to 32 labels per thread (B * L <= 32 768: every batch of this model) are requested together and kept in registers for
  // the second pass; the prefix sum is a wave scan + 16 wave totals (two barriers) — the loop of one load at a time and the
  // 20-barrier scan of round 4 took 39 us for 19 200 labels, between the uni-modal stacks and the joint pass
  constexpr int PER_MAX = 32;
  const bool fast = per <= PER_MAX;
  int64_t lab[PER_MAX];
  int c = 0, k, found;
  if (fast) {
#pragma unroll
    for (int j = 0; j < PER_MAX; ++j) lab[j] = (e0 + j < e1) ? labels[e0 + j] : (int64_t)-1;
#pragma unroll
    for (int j = 0; j < PER_MAX; ++j) c += lab[j] > -1 ? 1 : 0;
    const int lane = tid & 63, wave = tid >> 6;
    int incl = c;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int v = __shfl_up(incl, o);
      if (lane >= o) incl += v;
    }
    if (lane == 63) scan[wave] = incl;
    __syncthreads();
    int base = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < 16; ++w) {
      const int t = scan[w];
      base += (w < wave) ? t : 0;
      tot += t;
    }
    k = base + incl - c;
    found = tot;
  } else {
    for (int e = e0; e < e1; ++e) c += labels[e] > -1 ? 1 : 0;
    scan[tid] = c;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) {
      const int v = (tid >= o) ? scan[tid - o] : 0;
      __syncthreads();
      scan[tid] += v;
      __syncthreads();
    }
    k = scan[tid] - c;
    found = scan[1023];
  }
  if (found > n_out && tid == 0) {
    // fewer output slots than scored rows: the surplus drops out of the loss (a stale or wrong scored-row count in host_counts,
    // ADVICE r04).  More slots than rows is harmless (padded with -1 = ignored) and stays legal.  Nothing below goes out of
    // bounds, so the kernel does not trap (rounds 4-5 did: the whole process died where the reference raises catchably):
    // it reports through the caller's error word — the first error wins — and the host raises at its next count read-back.
    if (err != nullptr) {
      if (atomicCAS(err, 0ull, (unsigned long long)MVPTR_DEV_ERR_SCORED_ROWS) == 0ull) {
        err[1] = (unsigned long long)found;
        err[2] = (unsigned long long)n_out;
      }
    } else {
      printf("mvptr_compact_scored: %d scored rows but only %d output slots (host_counts.scored_* does not describe this batch)\n", found, n_out);
    }
  }
  if (fast) {
#pragma unroll
    for (int j = 0; j < PER_MAX; ++j) {
      const int64_t lb = lab[j];
      if (lb > -1) {
        if (k < n_out) {
          const int e = e0 + j;
          out_labels[k] = lb;
          const int b = e / L;
          out_rows[k] = pos ? pos[(int64_t)b * ld_pos + (e - b * L)] : e;
        }
        ++k;
      }
    }
  } else {
    for (int e = e0; e < e1; ++e) {
      const int64_t lb = labels[e];
      if (lb > -1) {
        if (k < n_out) {
          out_labels[k] = lb;
          const int b = e / L;
          out_rows[k] = pos ? pos[(int64_t)b * ld_pos + (e - b * L)] : e;
        }
        ++k;
      }
    }
  }
  for (int j = found + tid; j < n_out; j += 1024) {
    out_labels[j] = -1;
    out_rows[j] = -1;
  }
}
}  // namespace

extern "C" int mvptr_compact_scored(const int64_t* labels, const int32_t* pos, int64_t ld_pos, int B, int L, int n_out,
                                    int64_t* out_labels, int32_t* out_rows, int64_t* err, void* stream) {
  if (B <= 0 || L <= 0 || n_out < 0 || (int64_t)B * L >= (int64_t)1 << 30 || !labels || (n_out > 0 && (!out_labels || !out_rows)) ||
      (pos && ld_pos < L))
    MVPTR_FAIL(MVPTR_BAD_ARG, "compact_scored: bad argument");
  if (n_out == 0) return MVPTR_OK;
  hipLaunchKernelGGL(compact_scored_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, labels, pos, ld_pos, B, L, n_out, out_labels,
                     out_rows, reinterpret_cast<unsigned long long*>(err));
  MVPTR_CHECK_LAUNCH("compact_scored");
  return MVPTR_OK;
}

// Host-provided counts against the device's (sync-free training step: the input-only counts of a batch come from where
// the batch was built; the pack maps count them again on the device).  A mismatch is a broken caller contract, not bad data:
// every buffer, grid and LDS tile of the step has been sized from the host's numbers, so kernels queued behind this one would
// index past their buffers.  This is the one check that still TRAPS (round 6 turned the others into the device error word):
// stopping the queue is the only safe continuation.  model.verify_host_counts = True checks on the host instead (one
// read-back, catchable ValueError) while a data pipeline is being brought up.
namespace {
__global__ void check_counts_kernel(const int64_t* a, const int64_t* b, int64_t ra, int64_t la, int64_t rb, int64_t lb) {
  if (a[0] != ra || a[1] != la || b[0] != rb || b[1] != lb) {
    printf("mvptr_check_counts: host counts (%ld, %ld, %ld, %ld) != device counts (%ld, %ld, %ld, %ld)\n", (long)ra, (long)la, (long)rb,
           (long)lb, (long)a[0], (long)a[1], (long)b[0], (long)b[1]);
    __builtin_trap();
  }
}
}  // namespace

extern "C" int mvptr_check_counts(const int64_t* counts_a, const int64_t* counts_b, int64_t rows_a, int64_t lmax_a, int64_t rows_b,
                                  int64_t lmax_b, void* stream) {
  if (!counts_a || !counts_b) MVPTR_FAIL(MVPTR_BAD_ARG, "check_counts: NULL argument");
  hipLaunchKernelGGL(check_counts_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, counts_a, counts_b, rows_a, lmax_a, rows_b, lmax_b);
  MVPTR_CHECK_LAUNCH("check_counts");
  return MVPTR_OK;
}
