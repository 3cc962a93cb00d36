// diag.hip — measurement helpers (never on the product path).
//
// mvptr_diag_stream_read: reads `bytes` of a buffer exactly once through one of the two load paths
// the GEMM kernels use, so that rocprofv3's FETCH_SIZE can be calibrated against a KNOWN byte
// count in the same access pattern (MI355X_MICROARCH.md, HBM: the counter is uncalibrated outside
// 16-B-per-lane streaming reads).  mode 0: buffer_load_dwordx4 ... lds (LDS-DMA,
// 1 KiB per wave instruction, the operand path of gemm_nt / gemm_tn); mode 1: global_load_dwordx4
// to registers.  A checksum goes to `sink` so the reads stay live.
#include "common.h"

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

}  // namespace

extern "C" int mvptr_diag_stream_read(const void* src, int64_t bytes, int mode, float* sink, void* stream) {
  if (!src || !sink || bytes < 4096 || (bytes & 4095)) MVPTR_FAIL(MVPTR_BAD_ARG, "diag_stream_read: bytes must be a positive multiple of 4096");
  if (((uintptr_t)src & 15)) MVPTR_FAIL(MVPTR_BAD_ALIGN, "diag_stream_read: src must be 16-byte aligned");
  hipLaunchKernelGGL(stream_read_kernel, dim3(2048), dim3(256), 4096, (hipStream_t)stream, (const char*)src, bytes, mode, sink);
  MVPTR_CHECK_LAUNCH("diag_stream_read");
  return MVPTR_OK;
}
