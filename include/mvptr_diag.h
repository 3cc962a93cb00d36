/* mvptr_diag.h — measurement helpers of the DIAGNOSTIC build (libmvptr_hip_diag.so, `make -C mvp_pytorch_amd/csrc diag`).
 * Not part of the product ABI: libmvptr_hip.so does not export these symbols, mvp_pytorch_amd.hip binds them only when the
 * diagnostic library is loaded (MVPTR_LIB=diag, measurement tools under tools/). */
#ifndef MVPTR_DIAG_H
#define MVPTR_DIAG_H
#include "mvptr.h"
#ifdef __cplusplus
extern "C" {
#endif

/* Measurement helper (never on the product path): reads `bytes` (a multiple of 4096) of `src` exactly
 * once, mode 0 through buffer_load ... lds (the GEMM operand path), mode 1 through global_load_dwordx4,
 * so that rocprofv3's FETCH_SIZE can be calibrated against a known byte count (tools/calib_fetch.py). */
int mvptr_diag_stream_read(const void* src, int64_t bytes, int mode, float* sink, void* stream);
/* Measurement helper: per-CU store rate by access shape.  `blocks` 512-thread workgroups; every wave
 * instruction writes 1 KiB as rows_per_instr segments of 1024 / rows_per_instr bytes, `stride` bytes apart
 * (8 x 128 B at the output row stride = the GEMM epilogues' shape); tools/store_probe.py. */
int mvptr_diag_store_probe(void* dst, int64_t dst_bytes, int blocks, int64_t bytes_per_wave, int rows_per_instr,
                           int64_t stride, void* stream);
/* Measurement helper: operand-fill rate.  `blocks` 256-thread workgroups each stream their wg_bytes region
 * (all the same region when shared != 0) `reps` times in 32-KiB stages, three in flight — the GEMM
 * kernels' staging pattern alone.  mode 0: buffer_load ... lds, mode 1: buffer_load to registers;
 * the working-set size decides the level served from (L2 / Infinity Cache / HBM); tools/fill_probe.py. */
int mvptr_diag_fill_probe(const void* src, int64_t src_bytes, int blocks, int64_t wg_bytes, int reps, int shared, int mode,
                          float* sink, void* stream);

/* Kernel-configuration knobs of the diagnostic build (common.h MvptrKnobs): MVPTR_GEMM_CFG, MVPTR_GEMM_TN, MVPTR_NT_EXP, ... */
int mvptr_set_knob(const char* name, const char* value);

#ifdef __cplusplus
}
#endif
#endif /* MVPTR_DIAG_H */
