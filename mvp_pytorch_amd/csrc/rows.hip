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
// One workgroup: sequences are dealt to threads (count pass), a block-wide exclusive scan gives seq_start,
// a second pass writes pos_out[s, slot] = packed row (-1: padded slot) and idx_out[packed row] = source row.
__global__ __launch_bounds__(1024) void pack_maps_kernel(mvptr_pack_seg s0, mvptr_pack_seg s1, int nseg, int n_seq,
                                                          int32_t* pos_out, int32_t* idx_out, int32_t* seq_start,
                                                          int32_t* seq_len, int64_t* counts) {
  __shared__ int scan[1024];
  __shared__ int carry_s, maxlen_s;
  const int tid = threadIdx.x;
  const int Ltot = s0.len + (nseg > 1 ? s1.len : 0);
  if (tid == 0) {
    carry_s = 0;
    maxlen_s = 0;
  }
  __syncthreads();
  for (int base = 0; base < n_seq; base += 1024) {
    const int s = base + tid;
    int cnt = 0;
    if (s < n_seq) {
      {
        const int64_t r = s0.sel ? s0.sel[s] : (int64_t)s;
        const float* m = s0.mask + r * s0.ld_mask + s0.col0;
        for (int l = 0; l < s0.len; ++l) cnt += (m[l] == 0.f);
      }
      if (nseg > 1) {
        const int64_t r = s1.sel ? s1.sel[s] : (int64_t)s;
        const float* m = s1.mask + r * s1.ld_mask + s1.col0;
        for (int l = 0; l < s1.len; ++l) cnt += (m[l] == 0.f);
      }
    }
    // inclusive scan of cnt over the 1024 threads (Hillis-Steele in LDS)
    scan[tid] = cnt;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) {
      const int v = (tid >= o) ? scan[tid - o] : 0;
      __syncthreads();
      scan[tid] += v;
      __syncthreads();
    }
    const int carry = carry_s;
    int start = carry + scan[tid] - cnt;
    if (s < n_seq) {
      seq_start[s] = start;
      seq_len[s] = cnt;
      atomicMax(&maxlen_s, cnt);
      int32_t* po = pos_out + (int64_t)s * Ltot;
      {
        const int64_t r = s0.sel ? s0.sel[s] : (int64_t)s;
        const float* m = s0.mask + r * s0.ld_mask + s0.col0;
        for (int l = 0; l < s0.len; ++l) {
          const bool ok = (m[l] == 0.f);
          po[l] = ok ? start : -1;
          if (ok) {
            const int64_t src = s0.pos ? (int64_t)s0.pos[r * s0.ld_pos + s0.col0 + l] : r * s0.src_seq_stride + s0.col0 + l;
            idx_out[start++] = (int32_t)(src + s0.src_base);
          }
        }
      }
      if (nseg > 1) {
        const int64_t r = s1.sel ? s1.sel[s] : (int64_t)s;
        const float* m = s1.mask + r * s1.ld_mask + s1.col0;
        for (int l = 0; l < s1.len; ++l) {
          const bool ok = (m[l] == 0.f);
          po[s0.len + l] = ok ? start : -1;
          if (ok) {
            const int64_t src = s1.pos ? (int64_t)s1.pos[r * s1.ld_pos + s1.col0 + l] : r * s1.src_seq_stride + s1.col0 + l;
            idx_out[start++] = (int32_t)(src + s1.src_base);
          }
        }
      }
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
// dst[idx[i], :] += src[i, :] (bf16 destination, packed-pair atomics: rows may repeat and several calls may hit
// one row); src is bf16, or f32 (gradient of an f32 head, rounded to bf16 on the way) when src_f32
__global__ __launch_bounds__(256) void scatter_add_rows_kernel(const void* src, int64_t ld_src, int src_f32, const int32_t* idx,
                                                                __bf16* dst, int64_t ld_dst, __bf16* dst2, int64_t ld_dst2, int split,
                                                                int n, int pairs) {
  const int64_t total = (int64_t)n * pairs;
  for (int64_t e = blockIdx.x * (int64_t)256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
    const int r = (int)(e / pairs), c = (int)(e - (int64_t)r * pairs);
    const int d = idx[r];
    if (d < 0) continue;
    bf16x2 v;
    if (src_f32) {
      const float* sp = (const float*)src + (int64_t)r * ld_src + c * 2;
      v[0] = f2bf(sp[0]);
      v[1] = f2bf(sp[1]);
    } else {
      v = *reinterpret_cast<const bf16x2*>((const __bf16*)src + (int64_t)r * ld_src + c * 2);
    }
    __bf16* dp = (dst2 != nullptr && d >= split) ? dst2 + (int64_t)(d - split) * ld_dst2 + c * 2 : dst + (int64_t)d * ld_dst + c * 2;
    asm volatile("global_atomic_pk_add_bf16 %0, %1, off" : : "v"(dp), "v"(v) : "memory");
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
  hipLaunchKernelGGL(pack_maps_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, segs[0], s1, nseg, n_seq, pos_out, idx_out,
                     seq_start, seq_len, counts);
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
                                      int64_t ld_dst, void* dst2, int64_t ld_dst2, int split, int n, int H, void* stream) {
  if (n <= 0 || H <= 0 || (H & 1) || (ld_src & 1) || (ld_dst & 1) || (dst2 && (ld_dst2 & 1)))
    MVPTR_FAIL(MVPTR_BAD_SHAPE, "scatter_add_rows: H and the leading dimensions must be positive and even");
  if (!src || !idx || !dst || ((uintptr_t)src & 3) || ((uintptr_t)dst & 3) || ((uintptr_t)dst2 & 3))
    MVPTR_FAIL(MVPTR_BAD_ALIGN, "scatter_add_rows: NULL or unaligned pointer");
  const int64_t total = (int64_t)n * (H / 2);
  int grid = (int)((total + 255) / 256);
  if (grid > 16384) grid = 16384;
  hipLaunchKernelGGL(scatter_add_rows_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, src, ld_src, src_f32 ? 1 : 0, idx,
                     (__bf16*)dst, ld_dst, (__bf16*)dst2, ld_dst2, split, n, H / 2);
  MVPTR_CHECK_LAUNCH("scatter_add_rows");
  return MVPTR_OK;
}
