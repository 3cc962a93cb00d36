// pipeline.hip — device side of the input pipeline: region features arrive as the
// base64 text of the dataset's TSV rows and are decoded, truncated / zero-padded to R rows and cast
// on the GPU, so the host only copies text into pinned memory.
//
// Replaces OscarTSVDataset_C.get_img_feature oscar/oscar_datasets_ml/oscar_tsv4.py:696-724
//   feat = np.frombuffer(base64.b64decode(arr[-1]), np.float32).reshape(num_boxes, img_feature_dim)
// and the truncation / zero padding of __getitem__ :332-352 (rows beyond max_img_seq_length are
// dropped, missing rows are zero), plus data_process' images.to(dtype) run_pretrain_ml.py:501-504.
//
// Byte work, HBM bound: a thread owns 16 characters (one 16-byte load) = 12 bytes = 3 floats;
// consecutive threads own consecutive 16-byte groups (coalesced loads, 12-byte strided stores that
// the L2 merges into full lines).  Every sample's text starts on a 16-byte boundary of the staging
// buffer (the host packs it that way).
#include "common.h"

namespace {

// base64 alphabet (RFC 4648 §4, what base64.b64decode accepts) -> 6-bit value, -1 = not in the alphabet
__device__ __forceinline__ int b64_val(uint32_t c) {
  const uint32_t up = c - 'A', lo = c - 'a', dg = c - '0';
  int v = -1;
  v = (up < 26u) ? (int)up : v;
  v = (lo < 26u) ? (int)lo + 26 : v;
  v = (dg < 10u) ? (int)dg + 52 : v;
  v = (c == '+') ? 62 : v;
  v = (c == '/') ? 63 : v;
  return v;
}

// 4 characters packed little-endian in one dword -> 3 bytes (low 24 bits of the result, first
// byte in bits 0-7); `bad` collects characters outside the alphabet
__device__ __forceinline__ uint32_t b64_quad(uint32_t w, int& bad) {
  const int v0 = b64_val(w & 0xffu), v1 = b64_val((w >> 8) & 0xffu), v2 = b64_val((w >> 16) & 0xffu),
            v3 = b64_val(w >> 24);
  bad |= (v0 | v1 | v2 | v3) < 0;
  const uint32_t b = ((uint32_t)(v0 & 63) << 18) | ((uint32_t)(v1 & 63) << 12) | ((uint32_t)(v2 & 63) << 6) | (uint32_t)(v3 & 63);
  return ((b >> 16) & 0xffu) | (b & 0xff00u) | ((b & 0xffu) << 16);
}

__global__ __launch_bounds__(256) void b64_features_kernel(const uint8_t* __restrict__ text,
                                                            const int64_t* __restrict__ offsets,
                                                            const int64_t* __restrict__ n_chars,
                                                            const int32_t* __restrict__ num_boxes, int R, int D,
                                                            float* __restrict__ out_f32, __bf16* __restrict__ out_bf16,
                                                            int64_t ld_bf16, int32_t* __restrict__ err) {
  const int s = blockIdx.y;
  const int64_t t0 = offsets[s], t1 = t0 + n_chars[s];
  const int nb = num_boxes[s];
  const int64_t valid = (int64_t)min(max(nb, 0), R) * D;   // floats kept (rows beyond R are dropped)
  const int64_t total = (int64_t)R * D;
  // the text must hold the whole num_boxes x D array: 4 * ceil(4 nb D / 3) characters
  const int64_t need_chars = (((int64_t)max(nb, 0) * D * 4 + 2) / 3) * 4;
  const bool short_text = (nb < 0 || t1 - t0 < need_chars || (t0 & 15));
  if (threadIdx.x == 0 && blockIdx.x == 0 && short_text) atomicOr(err, 1);
  const int64_t groups = (total + 2) / 3;
  for (int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; g < groups; g += (int64_t)gridDim.x * blockDim.x) {
    const int64_t f0 = g * 3;
    float v[3] = {0.f, 0.f, 0.f};
    if (f0 < valid && t0 + g * 16 < t1) {   // the last group may read into the sample's 16-byte padding
      const u32x4 w = *reinterpret_cast<const u32x4*>(text + t0 + g * 16);
      int bad = 0;
      const uint32_t q0 = b64_quad(w[0], bad), q1 = b64_quad(w[1], bad), q2 = b64_quad(w[2], bad), q3 = b64_quad(w[3], bad);
      // 12 bytes, little-endian floats: q0 = bytes 0-2, q1 = 3-5, q2 = 6-8, q3 = 9-11
      const uint32_t d0 = q0 | (q1 << 24);
      const uint32_t d1 = (q1 >> 8) | (q2 << 16);
      const uint32_t d2 = (q2 >> 16) | (q3 << 8);
      v[0] = __builtin_bit_cast(float, d0);
      v[1] = __builtin_bit_cast(float, d1);
      v[2] = __builtin_bit_cast(float, d2);
      // only characters that feed kept floats and lie inside the sample's text are checked: the '='
      // padding, dropped rows and (for a text flagged short) the bytes past its end are not
      if (bad && !short_text) {   // a short text is an error already; its '=' padding sits among "kept" floats
        int kept_chars = 16;
        if (f0 + 2 >= valid) kept_chars = (int)(((valid - f0) * 16 + 2) / 3);   // 4 or 8 bytes -> 6 or 11 chars
        const int64_t inside = t1 - (t0 + g * 16);
        if (inside < kept_chars) kept_chars = (int)inside;
        if (kept_chars >= 16) {
          atomicOr(err, 2);
        } else {
          int b2 = 0;
          for (int c = 0; c < kept_chars; ++c) b2 |= b64_val(text[t0 + g * 16 + c]) < 0;
          if (b2) atomicOr(err, 2);
        }
      }
    }
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const int64_t f = f0 + j;
      if (f >= total) break;
      const float x = (f < valid) ? v[j] : 0.f;
      if (out_f32) out_f32[(int64_t)s * total + f] = x;
      if (out_bf16) {
        const int64_t row = f / D;
        const int col = (int)(f - row * D);
        __bf16* o = out_bf16 + ((int64_t)s * R + row) * ld_bf16;
        o[col] = f2bf(x);
        if (col == D - 1)
          for (int64_t c = D; c < ld_bf16; ++c) o[c] = f2bf(0.f);   // K padding of the embedding GEMM
      }
    }
  }
}

}  // namespace

extern "C" int mvptr_b64_decode_features(const void* text, const int64_t* offsets, const int64_t* n_chars,
                                         const int32_t* num_boxes,
                                         int n_samples, int R, int D, float* out_f32, void* out_bf16,
                                         int64_t ld_bf16, int32_t* err_flag, void* stream) {
  if (n_samples <= 0 || R <= 0 || D <= 0) MVPTR_FAIL(MVPTR_BAD_SHAPE, "b64_decode_features: n_samples, R, D must be > 0");
  if (!text || !offsets || !n_chars || !num_boxes || !err_flag) MVPTR_FAIL(MVPTR_BAD_ARG, "b64_decode_features: NULL argument");
  if (!out_f32 && !out_bf16) MVPTR_FAIL(MVPTR_BAD_ARG, "b64_decode_features: no output");
  if (out_bf16 && ld_bf16 < D) MVPTR_FAIL(MVPTR_BAD_SHAPE, "b64_decode_features: ld_bf16 < D");
  if ((uintptr_t)text & 15) MVPTR_FAIL(MVPTR_BAD_ALIGN, "b64_decode_features: text must be 16-byte aligned");
  const int64_t groups = ((int64_t)R * D + 2) / 3;
  int bx = (int)((groups + 255) / 256);
  if (bx > 1024) bx = 1024;
  hipLaunchKernelGGL(b64_features_kernel, dim3(bx, n_samples), dim3(256), 0, (hipStream_t)stream, (const uint8_t*)text,
                     offsets, n_chars, num_boxes, R, D, out_f32, (__bf16*)out_bf16, ld_bf16, err_flag);
  MVPTR_CHECK_LAUNCH("b64_decode_features");
  return MVPTR_OK;
}
