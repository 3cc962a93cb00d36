// diag.hip — measurement helpers (never on the product path): compiled into libmvptr_hip_diag.so only (`make diag`,
// -DMVPTR_DIAG_BUILD, declared in include/mvptr_diag.h); the product build of this file is empty.
//
// mvptr_diag_stream_read: reads `bytes` of a buffer exactly once through one of the two load paths
// the GEMM kernels use, so that rocprofv3's FETCH_SIZE can be calibrated against a KNOWN byte
// count in the same access pattern (MI355X_MICROARCH.md, HBM: the counter is uncalibrated outside
// 16-B-per-lane streaming reads).  mode 0: buffer_load_dwordx4 ... lds (LDS-DMA,
// 1 KiB per wave instruction, the operand path of gemm_nt / gemm_tn); mode 1: global_load_dwordx4
// to registers.  A checksum goes to `sink` so the reads stay live.
#include "common.h"
#ifdef MVPTR_DIAG_BUILD
#include "../../include/mvptr_diag.h"

namespace {

extern __shared__ __attribute__((aligned(1024))) char dlds[];

__global__ __launch_bounds__(256) void stream_read_kernel(const char* src, int64_t bytes, int mode, float* sink) {
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int64_t chunk = 4096;  // bytes per workgroup iteration: 4 waves x 1 KiB
  const int64_t nchunks = bytes / chunk;
  float acc = 0.f;
  if (mode == 0) {
    const uint32_t lds0 = lds_addr(dlds) + (uint32_t)wave * 1024u;
    for (int64_t c = blockIdx.x; c < nchunks; c += gridDim.x) {
      // a descriptor per chunk keeps the 32-bit buffer offset small for any buffer size
      const u32x4 rs = make_rsrc_words(src + c * chunk, (uint32_t)chunk);
      lds_dma16(rs, (uint32_t)(wave * 1024 + lane * 16), lds0);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      const f32x4 v = *reinterpret_cast<const f32x4*>(dlds + wave * 1024 + lane * 16);
      acc += v[0] + v[1] + v[2] + v[3];
    }
  } else {
    for (int64_t c = blockIdx.x; c < nchunks; c += gridDim.x) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(src + c * chunk + wave * 1024 + lane * 16);
      acc += v[0] + v[1] + v[2] + v[3];
    }
  }
  if (acc == 12345.678f) *sink = acc;
}

// Store-path probe: every wave instruction writes 1 KiB as R row segments of 1024 / R bytes, the rows
// `stride` bytes apart (R = 1: one contiguous KiB; R = 8: eight whole 128-byte lines, the GEMM
// epilogues' shape; R = 32: 32-byte segments).  grid = one 512-thread workgroup per CU-slot, each
// wave sweeps its own row range, so the figure is the per-CU store rate the epilogues see.
__global__ __launch_bounds__(512) void store_probe_kernel(char* dst, int64_t bytes_per_wave, int rows_per_instr, int64_t stride) {
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int seg = 1024 / rows_per_instr;            // bytes per row segment
  const int lanes_per_row = seg / 16;
  const int r = lane / lanes_per_row, c = lane % lanes_per_row;
  char* base = dst + ((int64_t)blockIdx.x * 8 + wave) * (bytes_per_wave / 1024) * rows_per_instr * stride;
  const u32x4 v = {(uint32_t)lane, 1u, 2u, 3u};
  const int64_t n = bytes_per_wave / 1024;
  for (int64_t i = 0; i < n; ++i)
    *reinterpret_cast<u32x4*>(base + (i * rows_per_instr + r) * stride + c * 16) = v;
}


// Operand-fill probe: every workgroup (256 threads, one per CU when the grid is the CU count) streams
// its `wg_bytes` region `reps` times in 32-KiB stages of 8 x 1 KiB per wave, three stages in flight
// (the GEMM kernels' staging pattern without their MFMAs, LDS reads and barriers).  mode 0: LDS-DMA
// (buffer_load_dwordx4 ... lds); mode 1: buffer_load_dwordx4 to registers.  shared != 0: all
// workgroups read the same region.  The working set (wg_bytes x grid, or wg_bytes when shared)
// decides whether the bytes come from L2, the Infinity Cache or HBM.
__global__ __launch_bounds__(256, 1) void fill_probe_kernel(const char* src, int64_t wg_bytes, int reps, int shared, int mode, float* sink) {
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const char* base = src + (shared ? (int64_t)0 : (int64_t)blockIdx.x * wg_bytes);
  const u32x4 rs = make_rsrc_words(base, (uint32_t)wg_bytes);
  const uint32_t lds0 = lds_addr(dlds) + (uint32_t)wave * 1024u;
  const int nst = (int)(wg_bytes >> 15);
  const int total = nst * reps;
  const uint32_t lane_off = (uint32_t)(wave * 1024 + lane * 16);
  if (mode == 0) {
    auto issue = [&](int s) {
      const uint32_t so = (s < total) ? (uint32_t)(s % nst) << 15 : MVPTR_OOB;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const uint32_t vo = (so == MVPTR_OOB) ? MVPTR_OOB : so + (uint32_t)i * 4096u + lane_off;
        lds_dma16(rs, vo, lds0 + (uint32_t)((s & 3) * 32768 + i * 4096));
      }
    };
    issue(0);
    issue(1);
    issue(2);
    for (int s = 0; s < total; ++s) {
      issue(s + 3);
      asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (reps < 0) *sink = *reinterpret_cast<const float*>(dlds + lane * 4);
  } else {
    const __amdgpu_buffer_rsrc_t rsb = make_rsrc(base, (uint32_t)wg_bytes);
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    u32x4 r[3][8];
    auto issue = [&](int s, u32x4(&d)[8]) {
      const uint32_t so = (s < total) ? (uint32_t)(s % nst) << 15 : MVPTR_OOB;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const uint32_t vo = (so == MVPTR_OOB) ? MVPTR_OOB : so + (uint32_t)i * 4096u + lane_off;
        d[i] = __builtin_amdgcn_raw_buffer_load_b128(rsb, vo, 0, 0);
      }
    };
    auto use = [&](u32x4(&d)[8]) {
#pragma unroll
      for (int i = 0; i < 8; ++i) acc[i & 3] += __builtin_bit_cast(float, d[i][i & 3]);
    };
    issue(0, r[0]);
    issue(1, r[1]);
    for (int s = 0; s < total; s += 3) {
      issue(s + 2, r[2]);
      use(r[0]);
      issue(s + 3, r[0]);
      use(r[1]);
      issue(s + 4, r[1]);
      use(r[2]);
    }
    if (acc[0] + acc[1] + acc[2] + acc[3] == 12345.678f) *sink = acc[0];
  }
}
}  // namespace

extern "C" int mvptr_diag_store_probe(void* dst, int64_t dst_bytes, int blocks, int64_t bytes_per_wave, int rows_per_instr,
                                      int64_t stride, void* stream) {
  if (!dst || blocks <= 0 || bytes_per_wave < 1024 || (bytes_per_wave & 1023)) MVPTR_FAIL(MVPTR_BAD_ARG, "diag_store_probe: bad argument");
  if (rows_per_instr < 1 || rows_per_instr > 64 || (64 % rows_per_instr) || stride < 1024 / rows_per_instr || (stride & 15))
    MVPTR_FAIL(MVPTR_BAD_ARG, "diag_store_probe: rows_per_instr must divide 64, stride >= segment");
  const int64_t need = (int64_t)blocks * 8 * (bytes_per_wave / 1024) * rows_per_instr * stride;
  if (need > dst_bytes) MVPTR_FAIL(MVPTR_BAD_ARG, "diag_store_probe: destination too small (%ld > %ld)", (long)need, (long)dst_bytes);
  hipLaunchKernelGGL(store_probe_kernel, dim3(blocks), dim3(512), 0, (hipStream_t)stream, (char*)dst, bytes_per_wave, rows_per_instr, stride);
  MVPTR_CHECK_LAUNCH("diag_store_probe");
  return MVPTR_OK;
}

extern "C" int mvptr_diag_stream_read(const void* src, int64_t bytes, int mode, float* sink, void* stream) {
  if (!src || !sink || bytes < 4096 || (bytes & 4095)) MVPTR_FAIL(MVPTR_BAD_ARG, "diag_stream_read: bytes must be a positive multiple of 4096");
  if (((uintptr_t)src & 15)) MVPTR_FAIL(MVPTR_BAD_ALIGN, "diag_stream_read: src must be 16-byte aligned");
  hipLaunchKernelGGL(stream_read_kernel, dim3(2048), dim3(256), 4096, (hipStream_t)stream, (const char*)src, bytes, mode, sink);
  MVPTR_CHECK_LAUNCH("diag_stream_read");
  return MVPTR_OK;
}

extern "C" int mvptr_diag_fill_probe(const void* src, int64_t src_bytes, int blocks, int64_t wg_bytes, int reps, int shared, int mode,
                                     float* sink, void* stream) {
  if (!src || !sink || blocks <= 0 || reps <= 0 || wg_bytes < 32768 || (wg_bytes & 32767) || wg_bytes >= ((int64_t)1 << 31))
    MVPTR_FAIL(MVPTR_BAD_ARG, "diag_fill_probe: wg_bytes must be a multiple of 32 KiB below 2 GiB, blocks and reps > 0");
  if ((shared ? wg_bytes : wg_bytes * blocks) > src_bytes) MVPTR_FAIL(MVPTR_BAD_ARG, "diag_fill_probe: source too small");
  if (((uintptr_t)src & 15)) MVPTR_FAIL(MVPTR_BAD_ALIGN, "diag_fill_probe: src must be 16-byte aligned");
  hipError_t e = hipFuncSetAttribute((const void*)fill_probe_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
  if (e != hipSuccess) MVPTR_FAIL(MVPTR_HIP_ERROR, "diag_fill_probe: set LDS size: %s", hipGetErrorString(e));
  hipLaunchKernelGGL(fill_probe_kernel, dim3(blocks), dim3(256), 131072, (hipStream_t)stream, (const char*)src, wg_bytes, reps, shared,
                     mode, sink);
  MVPTR_CHECK_LAUNCH("diag_fill_probe");
  return MVPTR_OK;
}
#endif  // MVPTR_DIAG_BUILD
