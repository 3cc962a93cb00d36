// gemm_nt.hip — C[M,N] = A[M,K] * B[N,K]^T (bf16 in, f32 accumulate) with fused epilogues.
//
// Reference op sequences replaced: nn.Linear forward + the elementwise ops that follow it in
// transformers/pytorch_transformers/modeling_bert.py:348-352 (dense+dropout+residual),
// :394-397 (dense+gelu :142-148), :407-411, oscar/modeling/modeling_vlbert.py:71-73 (Q/K/V),
// and the data-gradient GEMMs autograd derives from them.
//
// CDNA4 design: 128x128x64 tile per 256-thread workgroup (4 waves as 2x2, each 64x64 =
// 4x4 v_mfma_f32_16x16x32_bf16 tiles), operands staged HBM -> LDS with buffer_load ... lds
// (16 B per lane, no VGPR round trip), double buffered, XOR-swizzled 128-B rows so every
// ds_read_b128 fragment read is bank-conflict free, XCD-aware tile order.  Out-of-range
// rows / K tail come back as zeros from the buffer bounds check.
// The MFMA is issued with the weight tile as the A operand and the activation tile as the
// B operand, so a lane ends up with 4 consecutive output columns of one row (8-byte stores).
#include "common.h"

namespace {

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int TILE_BYTES = BM * BK * 2;  // 16 KiB per operand per stage

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
  int vec_store;  // ldc % 4 == 0 and 8-byte aligned bases: packed stores allowed
};

__device__ __forceinline__ void store_bf16x4(__bf16* base, int64_t ld, int m, int n, int N,
                                             bool vec, const float v[4]) {
  __bf16* p = base + (int64_t)m * ld + n;
  if (vec && n + 3 < N) {
    bf16x4 o = {f2bf(v[0]), f2bf(v[1]), f2bf(v[2]), f2bf(v[3])};
    *reinterpret_cast<bf16x4*>(p) = o;
  } else {
#pragma unroll
    for (int r = 0; r < 4; ++r)
      if (n + r < N) p[r] = f2bf(v[r]);
  }
}

__device__ __forceinline__ void load_bf16x4(const __bf16* base, int64_t ld, int m, int n, int N,
                                            bool vec, float v[4]) {
  const __bf16* p = base + (int64_t)m * ld + n;
  if (vec && n + 3 < N) {
    bf16x4 o = *reinterpret_cast<const bf16x4*>(p);
#pragma unroll
    for (int r = 0; r < 4; ++r) v[r] = bf2f(o[r]);
  } else {
#pragma unroll
    for (int r = 0; r < 4; ++r) v[r] = (n + r < N) ? bf2f(p[r]) : 0.f;
  }
}

template <int EPI>
__global__ __launch_bounds__(256) void gemm_nt_kernel(GemmNtArgs p) {
  __shared__ __attribute__((aligned(1024))) char lds[4 * TILE_BYTES];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nwg = p.tiles_m * p.tiles_n;
  const int t = xcd_remap(blockIdx.x, nwg);
  const int tm = t / p.tiles_n;
  const int tn = t - tm * p.tiles_n;
  const int m0 = tm * BM, n0 = tn * BN;
  const int rows_a = min(BM, p.M - m0);
  const int rows_b = min(BN, p.N - n0);

  const __amdgpu_buffer_rsrc_t rsA =
      make_rsrc(p.A + (int64_t)m0 * p.lda, (uint32_t)(((int64_t)(rows_a - 1) * p.lda + p.K) * 2));
  const __amdgpu_buffer_rsrc_t rsB =
      make_rsrc(p.B + (int64_t)n0 * p.ldb, (uint32_t)(((int64_t)(rows_b - 1) * p.ldb + p.K) * 2));

  // staging: instruction i of this wave fills LDS rows (i*4+wave)*8 .. +8 of a tile
  // (1 KiB, lane-linear); lane -> (row, physical chunk); logical chunk = phys ^ swz(row)
  uint32_t offA[4], offB[4];
  int kc[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = (i * 4 + wave) * 8 + (lane >> 3);
    const int c = (lane & 7) ^ ((row >> 1) & 7);
    kc[i] = c * 8;
    offA[i] = (uint32_t)(row * p.lda * 2 + c * 16);
    offB[i] = (uint32_t)(row * p.ldb * 2 + c * 16);
  }
  auto stage = [&](int buf, int k0) {
    char* la = lds + buf * 2 * TILE_BYTES;
    char* lb = la + TILE_BYTES;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const bool kok = (k0 + kc[i]) < p.K;
      const uint32_t va = kok ? offA[i] + (uint32_t)k0 * 2 : MVPTR_OOB;
      const uint32_t vb = kok ? offB[i] + (uint32_t)k0 * 2 : MVPTR_OOB;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, LDS_PTR(la + (i * 4 + wave) * 1024), 16, va,
                                               0, 0, 0);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, LDS_PTR(lb + (i * 4 + wave) * 1024), 16, vb,
                                               0, 0, 0);
    }
  };

  const int wm = wave >> 1, wn = wave & 1;
  const int c16 = lane & 15, q4 = lane >> 4;
  // fragment read byte offsets inside a tile (per mt / nt, ks)
  uint32_t fx[4][2], fw[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int rx = wm * 64 + i * 16 + c16;
    const int rw = wn * 64 + i * 16 + c16;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int ch = ks * 4 + q4;
      fx[i][ks] = rx * 128 + ((ch ^ ((rx >> 1) & 7)) << 4);
      fw[i][ks] = rw * 128 + ((ch ^ ((rw >> 1) & 7)) << 4);
    }
  }

  f32x4 acc[4][4];  // [nt][mt]
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int nk = (p.K + BK - 1) / BK;
  stage(0, 0);
  for (int kt = 0; kt < nk; ++kt) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (kt + 1 < nk) stage((kt + 1) & 1, (kt + 1) * BK);
    const char* la = lds + (kt & 1) * 2 * TILE_BYTES;
    const char* lb = la + TILE_BYTES;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8 xf[4], wf[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        xf[i] = *reinterpret_cast<const bf16x8*>(la + fx[i][ks]);
        wf[i] = *reinterpret_cast<const bf16x8*>(lb + fw[i][ks]);
      }
#pragma unroll
      for (int nt = 0; nt < 4; ++nt)
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
          acc[nt][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[nt], xf[mt], acc[nt][mt], 0, 0, 0);
    }
  }

  // ------------------------------------------------------------------ epilogue
  const bool vec = p.vec_store != 0;
#pragma unroll
  for (int nt = 0; nt < 4; ++nt) {
    const int n = n0 + wn * 64 + nt * 16 + q4 * 4;
    float b4[4] = {0.f, 0.f, 0.f, 0.f};
    if (EPI != MVPTR_EPI_GELU_BWD && EPI != MVPTR_EPI_ADD) {
      if (p.bias != nullptr) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (n + r < p.N) b4[r] = p.bias[n + r];
      }
    }
    float cs[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
      const int m = m0 + wm * 64 + mt * 16 + c16;
      const bool mok = (m < p.M) && (n < p.N);
      float v[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] = acc[nt][mt][r] + b4[r];
      if (EPI == MVPTR_EPI_BIAS) {
        if (mok) store_bf16x4((__bf16*)p.out0, p.ldc, m, n, p.N, vec, v);
      } else if (EPI == MVPTR_EPI_BIAS_GELU) {
        if (mok) {
          store_bf16x4((__bf16*)p.out0, p.ldc, m, n, p.N, vec, v);
          float g[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) g[r] = gelu_erf(bf2f(f2bf(v[r])));
          store_bf16x4((__bf16*)p.out1, p.ldc, m, n, p.N, vec, g);
        }
      } else if (EPI == MVPTR_EPI_BIAS_RESID) {
        if (mok) {
          float a[4];
          load_bf16x4(p.aux, p.ld_aux, m, n, p.N, vec, a);
#pragma unroll
          for (int r = 0; r < 4; ++r)
            v[r] = drop_apply(p.drop, (uint64_t)m * (uint64_t)p.N + (uint64_t)(n + r), v[r]) + a[r];
          store_bf16x4((__bf16*)p.out0, p.ldc, m, n, p.N, vec, v);
        }
      } else if (EPI == MVPTR_EPI_GELU_BWD) {
        if (mok) {
          float a[4];
          load_bf16x4(p.aux, p.ld_aux, m, n, p.N, vec, a);
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            v[r] = bf2f(f2bf(v[r] * gelu_erf_grad(a[r])));
            cs[r] += v[r];
          }
          store_bf16x4((__bf16*)p.out0, p.ldc, m, n, p.N, vec, v);
        }
      } else if (EPI == MVPTR_EPI_ADD) {
        if (mok) {
          if (p.aux != nullptr) {
            float a[4];
            load_bf16x4(p.aux, p.ld_aux, m, n, p.N, vec, a);
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] += a[r];
          }
          store_bf16x4((__bf16*)p.out0, p.ldc, m, n, p.N, vec, v);
        }
      } else if (EPI == MVPTR_EPI_F32) {
        if (mok) {
          float* o = (float*)p.out0 + (int64_t)m * p.ldc + n;
          if (vec && n + 3 < p.N) {
            *reinterpret_cast<f32x4*>(o) = f32x4{v[0], v[1], v[2], v[3]};
          } else {
#pragma unroll
            for (int r = 0; r < 4; ++r)
              if (n + r < p.N) o[r] = v[r];
          }
        }
      } else if (EPI == MVPTR_EPI_BIAS_TANH) {
        if (mok) {
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] = tanhf(v[r]);
          store_bf16x4((__bf16*)p.out0, p.ldc, m, n, p.N, vec, v);
        }
      }
    }
    if (EPI == MVPTR_EPI_GELU_BWD && p.vec_out != nullptr) {
      // reduce the 16 rows held by lanes with equal q4 (xor over the low 4 lane bits)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float s = cs[r];
        s += __shfl_xor(s, 1);
        s += __shfl_xor(s, 2);
        s += __shfl_xor(s, 4);
        s += __shfl_xor(s, 8);
        if (c16 == 0 && n + r < p.N) atomicAdd(p.vec_out + n + r, s);
      }
    }
  }
}

template <int EPI>
int launch(const GemmNtArgs& a, hipStream_t s) {
  const int nwg = a.tiles_m * a.tiles_n;
  hipLaunchKernelGGL(gemm_nt_kernel<EPI>, dim3(nwg), dim3(256), 0, s, a);
  MVPTR_CHECK_LAUNCH("gemm_nt");
  return MVPTR_OK;
}

}  // namespace

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
  if ((int64_t)128 * lda * 2 >= (int64_t)0x7fffffff || (int64_t)128 * ldb * 2 >= (int64_t)0x7fffffff)
    MVPTR_FAIL(MVPTR_BAD_SHAPE, "gemm_nt: leading dimension too large");
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
  a.tiles_m = (M + BM - 1) / BM;
  a.tiles_n = (N + BN - 1) / BN;
  const int esz = (epilogue == MVPTR_EPI_F32) ? 4 : 2;
  bool vec = (ldc % 4 == 0) && (((uintptr_t)out0 % (4 * esz)) == 0);
  if (out1) vec = vec && (((uintptr_t)out1 & 7) == 0);
  if (aux) vec = vec && (ld_aux % 4 == 0) && (((uintptr_t)aux & 7) == 0);
  a.vec_store = vec ? 1 : 0;
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
