"""ctypes binding of libmvptr_hip.so (the C ABI declared in include/mvptr.h).

PyTorch is used here only for device memory and streams: every wrapper takes torch tensors,
passes raw device pointers + the current HIP stream, and raises ``RuntimeError`` with
``mvptr_last_error()`` on failure (the reference raises ValueError/RuntimeError from the same
places, e.g. oscar/modeling/modeling_vlbert.py:435,542).  There is no CPU fallback: importing
works without a GPU (so host logic is testable), calling a kernel without the library or a
device fails loudly.
"""
import ctypes
import os
from ctypes import POINTER, Structure, c_char_p, c_float, c_int, c_int64, c_uint32, c_uint64, c_void_p

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# MVPTR_LIB=<tag> (measurement tools only): libmvptr_hip_<tag>.so instead of the product library — "diag" = the
# diagnostic build with kernel-configuration knobs (`make diag`); A/B runs load a second build of the library this way
LIB_PATH = os.path.join(_HERE, "csrc", "libmvptr_hip_%s.so" % os.environ["MVPTR_LIB"] if os.environ.get("MVPTR_LIB") else "libmvptr_hip.so")

# epilogue codes (mvptr_epilogue)
EPI_BIAS, EPI_BIAS_GELU, EPI_BIAS_RESID, EPI_GELU_BWD, EPI_ADD, EPI_F32, EPI_BIAS_TANH, EPI_BIAS_GELU_BF16, EPI_GELU_BWD_BF16 = range(9)

# every symbol include/mvptr.h declares (checked by tests/test_abi.py)
SYMBOLS = [
    "mvptr_query", "mvptr_last_error", "mvptr_gemm_nt", "mvptr_gemm_tn", "mvptr_gemm_tn_multi", "mvptr_colsum",
    "mvptr_attention_fwd", "mvptr_attention_bwd", "mvptr_attention_fwd_packed", "mvptr_attention_bwd_packed", "mvptr_attention_probs", "mvptr_layernorm_fwd", "mvptr_layernorm_bwd",
    "mvptr_layernorm_bwd_ws_bytes", "mvptr_embed_fwd", "mvptr_embed_bwd", "mvptr_cast_pack", "mvptr_cast_multi", "mvptr_cast_f32", "mvptr_ce_fwd",
    "mvptr_ce_bwd", "mvptr_adamw_multi", "mvptr_dropout_mask", "mvptr_layer_saved_bytes", "mvptr_layer_workspace_bytes",
    "mvptr_encoder_layer_fwd", "mvptr_encoder_layer_bwd", "mvptr_b64_decode_features",
    "mvptr_decoder_ce_fwd", "mvptr_decoder_ce_bwd",
    "mvptr_adamw_mirror_multi", "mvptr_sumsq_partials", "mvptr_sumsq_partial", "mvptr_clip_coef",
    "mvptr_sgemm_small", "mvptr_l2norm_fwd", "mvptr_l2norm_bwd", "mvptr_clip_ce_fwd", "mvptr_clip_ce_bwd",
    "mvptr_gather_rows", "mvptr_scatter_add_rows", "mvptr_ce_mean_small", "mvptr_pack_maps", "mvptr_gemm_nt_splitk",
    "mvptr_gemm_nt_ln", "mvptr_ln_stats_finalize",
    "mvptr_wra_rows", "mvptr_wra_fwd", "mvptr_wra_bwd", "mvptr_gemm_tn_multi_ws", "mvptr_gemm_tn_ws_bytes",
    "mvptr_hard_negative_mine", "mvptr_bce_logits", "mvptr_check_counts", "mvptr_tap_rows_bwd",
    "mvptr_masked_mean", "mvptr_dgelu_mul", "mvptr_compact_scored",
    "mvptr_gemm_tn_stack", "mvptr_encoder_layer_bwd_defer", "mvptr_set_dropout_salt",
]
# include/mvptr_diag.h: exported by the diagnostic build only (MVPTR_LIB=diag)
DIAG_SYMBOLS = ["mvptr_diag_stream_read", "mvptr_diag_store_probe", "mvptr_diag_fill_probe", "mvptr_set_knob"]


class Dropout(Structure):
    _fields_ = [("seed_lo", c_uint32), ("seed_hi", c_uint32), ("thresh16", c_uint32), ("pad_", c_uint32)]


class LayerDesc(Structure):
    _fields_ = [("B", c_int), ("L", c_int), ("H", c_int), ("heads", c_int), ("I", c_int),
                ("eps", c_float), ("training", c_int), ("p_hidden16", c_uint32),
                ("p_attn16", c_uint32), ("seed", c_uint64),
                ("M", c_int), ("M_plan", c_int), ("seq_start", c_void_p), ("seq_len", c_void_p), ("rows_dev", c_void_p),
                ("stash_bf16", c_int), ("beside", c_int)]


class LayerWeights(Structure):
    _fields_ = [(n, c_void_p) for n in (
        "w_qkv", "w_qkv_t", "b_qkv", "w_o", "w_o_t", "b_o", "ln1_g", "ln1_b", "w_i", "w_i_t", "b_i",
        "w_out", "w_out_t", "b_out", "ln2_g", "ln2_b")]


class TnProblem(Structure):
    _fields_ = [("A", c_void_p), ("lda", c_int64), ("B", c_void_p), ("ldb", c_int64), ("M", c_int), ("N", c_int),
                ("K", c_int), ("dW", c_void_p), ("ldw", c_int64), ("colsum", c_void_p)]


class PackSeg(Structure):
    _fields_ = [("mask", c_void_p), ("ld_mask", c_int64), ("sel", c_void_p), ("col0", c_int), ("len", c_int),
                ("pos", c_void_p), ("ld_pos", c_int64), ("src_seq_stride", c_int64), ("src_base", c_int64)]


class Tap(Structure):      # mvptr_tap
    _fields_ = [("g", c_void_p), ("ld_g", c_int64), ("idx", c_void_p), ("n", c_int), ("g_f32", c_int)]


TAP_MAX = 12


class LayerGrads(Structure):
    _fields_ = [(n, c_void_p) for n in (
        "w_qkv", "b_qkv", "w_o", "b_o", "ln1_g", "ln1_b", "w_i", "b_i", "w_out", "b_out", "ln2_g",
        "ln2_b")]


_lib = None


def load():
    """Load the shared library (once).  Raises if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            "libmvptr_hip.so is not built: run `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C mvp_pytorch_amd/csrc` (expected at %s)" % LIB_PATH)
    lib = ctypes.CDLL(LIB_PATH)
    lib.mvptr_last_error.restype = c_char_p
    lib.mvptr_layer_saved_bytes.restype = c_int64
    lib.mvptr_layer_workspace_bytes.restype = c_int64
    lib.mvptr_layer_saved_bytes.argtypes = [POINTER(LayerDesc)]
    lib.mvptr_layer_workspace_bytes.argtypes = [POINTER(LayerDesc)]
    P, I64, I, F = c_void_p, c_int64, c_int, c_float
    lib.mvptr_query.argtypes = [I, POINTER(c_int64)]
    if hasattr(lib, "mvptr_set_knob"):      # diagnostic build only
        lib.mvptr_set_knob.argtypes = [c_char_p, c_char_p]
    lib.mvptr_gemm_nt.argtypes = [P, I64, P, I64, I, I, I, I, P, P, I64, P, P, I64, P, POINTER(Dropout), P]
    lib.mvptr_gemm_nt_splitk.argtypes = [P, I64, P, I64, I, I, I, I, P, I64, P]
    lib.mvptr_gemm_nt_ln.argtypes = [P, I64, P, I64, I, I, I, I, P, P, I64, P, P, P, P, P, I64, P, P]
    lib.mvptr_ln_stats_finalize.argtypes = [P, I, I, c_float, P, P]
    lib.mvptr_gemm_tn.argtypes = [P, I64, P, I64, I, I, I, P, I64, P, P]
    lib.mvptr_gemm_tn_multi.argtypes = [POINTER(TnProblem), I, P]
    lib.mvptr_gemm_tn_multi_ws.argtypes = [POINTER(TnProblem), I, P, I64, P]
    lib.mvptr_gemm_tn_ws_bytes.restype = c_int64
    lib.mvptr_gemm_tn_ws_bytes.argtypes = [POINTER(TnProblem), I]
    lib.mvptr_colsum.argtypes = [P, I64, I, I, P, P]
    lib.mvptr_cast_multi.argtypes = [P, P, I, I, P]
    lib.mvptr_attention_fwd.argtypes = [P, P, P, P, I, I, I, POINTER(Dropout), P]
    lib.mvptr_attention_bwd.argtypes = [P, P, P, P, P, P, I, I, I, POINTER(Dropout), P]
    lib.mvptr_attention_probs.argtypes = [P, P, P, I, I, I, P]
    lib.mvptr_attention_fwd_packed.argtypes = [P, P, P, P, P, P, I, I, I, POINTER(Dropout), P]
    lib.mvptr_attention_bwd_packed.argtypes = [P, P, P, P, P, P, P, P, I, I, I, POINTER(Dropout), P]
    lib.mvptr_layernorm_fwd.argtypes = [P, P, P, F, P, P, P, I, I, I, I, I, POINTER(Dropout), P]
    lib.mvptr_layernorm_bwd.argtypes = [P, P, P, P, P, P, P, P, P, P, I, I, I, I, I, POINTER(Dropout), POINTER(Dropout), P, I64, P]
    lib.mvptr_layernorm_bwd_ws_bytes.restype = c_int64
    lib.mvptr_layernorm_bwd_ws_bytes.argtypes = [I, I]
    lib.mvptr_embed_fwd.argtypes = [P, P, P, P, P, P, P, I, I, I64, I64, I64, P]
    lib.mvptr_embed_bwd.argtypes = [P, P, P, P, P, P, P, I, I, P]
    lib.mvptr_cast_pack.argtypes = [P, I64, I, I, P, I64, P, I64, I, P]
    lib.mvptr_cast_f32.argtypes = [P, I64, I, I, P, I64, P]
    lib.mvptr_ce_fwd.argtypes = [P, I64, P, P, P, I, I, P]
    lib.mvptr_ce_bwd.argtypes = [P, I64, P, P, P, P, I64, I, I, I, P]
    lib.mvptr_dropout_mask.argtypes = [POINTER(Dropout), I64, P, P]
    if hasattr(lib, "mvptr_set_dropout_salt"):      # (absent from pre-ABI-7 builds loaded for A/B runs through MVPTR_LIB)
        lib.mvptr_set_dropout_salt.argtypes = [P]
    lib.mvptr_adamw_multi.argtypes = [P, P, P, I, I, F, F, F, P, P]
    lib.mvptr_adamw_mirror_multi.argtypes = [P, P, I, I, F, F, F, P, P]
    lib.mvptr_sumsq_partials.restype = c_int64
    lib.mvptr_sumsq_partials.argtypes = [I64]
    lib.mvptr_sumsq_partial.argtypes = [P, I64, P, P]
    lib.mvptr_clip_coef.argtypes = [P, I, F, P, P, P]
    lib.mvptr_sgemm_small.argtypes = [P, I64, I, I, P, P, I64, I, I, P, I, I, I, F, P, I, I, P, I64, P]
    lib.mvptr_ce_mean_small.argtypes = [P, I64, P, I, I, P, P, P]
    lib.mvptr_hard_negative_mine.argtypes = [P, I, I64, P, P, P, P, P, P, P, P]
    lib.mvptr_bce_logits.argtypes = [P, P, I, I, P, P, P, I, P]
    lib.mvptr_check_counts.argtypes = [P, P, I64, I64, I64, I64, P]
    lib.mvptr_masked_mean.argtypes = [P, P, I, P, P]
    lib.mvptr_compact_scored.argtypes = [P, P, I64, I, I, I, P, P, P, P]
    lib.mvptr_dgelu_mul.argtypes = [P, I64, P, I64, I, P, I64, I, I, I, P]
    lib.mvptr_tap_rows_bwd.argtypes = [POINTER(Tap), I, P, I64, I, P, I64, I, I, P, I64, P]
    lib.mvptr_l2norm_fwd.argtypes = [P, P, P, I, I, F, P]
    lib.mvptr_l2norm_bwd.argtypes = [P, P, P, P, I, I, P]
    lib.mvptr_clip_ce_fwd.argtypes = [P, I, I64, P, P, P, P, P]
    lib.mvptr_clip_ce_bwd.argtypes = [P, I, I64, P, P, P, P, P, P, P]
    lib.mvptr_gather_rows.argtypes = [P, I64, P, I64, I, P, P, I64, I, I, P]
    lib.mvptr_scatter_add_rows.argtypes = [P, I64, I, P, P, I64, P, I64, I, I, I, I, P]
    lib.mvptr_pack_maps.argtypes = [POINTER(PackSeg), I, I, P, P, P, P, P, P]
    lib.mvptr_wra_rows.argtypes = [P, I, P, P, I, I, I, P, P, P]
    lib.mvptr_wra_fwd.argtypes = [P, P, P, P, P, P, P, I, I, I, I, P, P, P, P, P, P, P, P, P]
    lib.mvptr_wra_bwd.argtypes = [P, P, P, I, I, I, I, P, P, P, P, P, P, P, P, P, P]
    lib.mvptr_b64_decode_features.argtypes = [P, P, P, P, I, I, I, P, P, I64, P, P]
    if hasattr(lib, "mvptr_diag_stream_read"):      # diagnostic build only (include/mvptr_diag.h)
        lib.mvptr_diag_stream_read.argtypes = [P, I64, I, P, P]
        lib.mvptr_diag_store_probe.argtypes = [P, I64, I, I64, I, I64, P]
        lib.mvptr_diag_fill_probe.argtypes = [P, I64, I, I64, I, I, I, P, P]
    lib.mvptr_decoder_ce_fwd.argtypes = [P, I64, P, I64, P, P, I, I, I, P, P, P, P, P]
    lib.mvptr_decoder_ce_bwd.argtypes = [P, I64, P, I64, P, P, P, P, I, I, I, P, I64, I, P]
    lib.mvptr_encoder_layer_fwd.argtypes = [POINTER(LayerDesc), POINTER(LayerWeights), P, P, P, P, P, I64, P]
    lib.mvptr_encoder_layer_bwd.argtypes = [POINTER(LayerDesc), POINTER(LayerWeights), P, P, P, P, P, POINTER(LayerGrads), P, I64, P]
    lib.mvptr_encoder_layer_bwd_defer.argtypes = [POINTER(LayerDesc), POINTER(LayerWeights), P, P, P, P, P, POINTER(LayerGrads), P, I64,
                                                  POINTER(TnProblem), POINTER(c_int), P]
    lib.mvptr_gemm_tn_stack.argtypes = [POINTER(TnProblem), I, P, I, P]
    _lib = lib
    return lib


def _check(rc):
    if rc != 0:
        raise RuntimeError("mvptr error %d: %s" % (rc, load().mvptr_last_error().decode()))


def _stream():
    return c_void_p(torch.cuda.current_stream().cuda_stream)


def _p(t):
    if t is None:
        return None
    if not t.is_cuda:
        raise RuntimeError("mvp_pytorch_amd: tensor is not on a HIP device — the encoder path has no CPU fallback")
    return c_void_p(t.data_ptr())


def make_dropout(p, seed):
    """Dropout descriptor for probability p and a 64-bit seed (None when p == 0)."""
    t = int(round(float(p) * 65536.0))
    if t <= 0:
        return None
    if t >= 65536:
        raise ValueError("dropout p must be < 1")
    return Dropout(seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF, t, 0)


def _dp(d):
    return ctypes.byref(d) if d is not None else None


def set_knob(name, value=""):
    """Diagnostic override of a kernel-configuration knob (tools/ only; see mvptr.h)."""
    _check(load().mvptr_set_knob(name.encode(), str(value).encode()))


def query(what):
    out = c_int64(0)
    _check(load().mvptr_query(what, ctypes.byref(out)))
    return out.value


# ------------------------------------------------------------------------------------------ ops
def gemm_nt(a, b, epilogue=EPI_BIAS, bias=None, aux=None, out=None, out1=None, vec_out=None,
            drop=None, n=None):
    """out[M,N] = a[M,K] @ b[N,K]^T with a fused epilogue.  a, b: bf16 2-D (row stride arbitrary)."""
    assert a.dtype == torch.bfloat16 and b.dtype == torch.bfloat16
    assert a.dim() == 2 and b.dim() == 2 and a.stride(1) == 1 and b.stride(1) == 1
    M, K = a.shape
    N = b.shape[0] if n is None else n
    assert b.shape[1] == K
    if out is None:
        # EPI_BIAS_GELU: out = the 8-bit gelu'(u) stash (dgelu_decode), out1 = gelu(u) in bf16
        dt = torch.float32 if epilogue == EPI_F32 else (torch.uint8 if epilogue == EPI_BIAS_GELU else torch.bfloat16)
        out = torch.empty((M, N), device=a.device, dtype=dt)
    if epilogue in (EPI_BIAS_GELU, EPI_BIAS_GELU_BF16):
        assert out.dtype == (torch.uint8 if epilogue == EPI_BIAS_GELU else torch.bfloat16)
        if out1 is None:
            out1 = torch.empty((M, N), device=a.device, dtype=torch.bfloat16)
        assert out1.stride(0) == out.stride(0)
    if epilogue == EPI_GELU_BWD:
        assert aux is not None and aux.dtype == torch.uint8
    if epilogue == EPI_GELU_BWD_BF16:
        assert aux is not None and aux.dtype == torch.bfloat16
    assert out.stride(1) == 1
    _check(load().mvptr_gemm_nt(_p(a), a.stride(0), _p(b), b.stride(0), M, N, K, epilogue, _p(bias),
                                _p(aux), aux.stride(0) if aux is not None else 0, _p(out), _p(out1),
                                out.stride(0), _p(vec_out), _dp(drop), _stream()))
    return (out, out1) if epilogue in (EPI_BIAS_GELU, EPI_BIAS_GELU_BF16) else out


DGELU_SCALE, DGELU_ZERO = 200.0, 27.0


def dgelu_decode(q):
    """The 8-bit gelu'(u) stash of EPI_BIAS_GELU (csrc/common.h: q = rint(200 g) + 26) -> f32."""
    return (q.to(torch.float32) - DGELU_ZERO) * (1.0 / DGELU_SCALE)


def dgelu_encode(g):
    """f32 gelu' values -> the 8-bit stash (test support; the kernels write it themselves)."""
    return torch.clamp(torch.round(g.float() * DGELU_SCALE) + DGELU_ZERO, 0, 255).to(torch.uint8)


LN_FOLD_BIAS, LN_FOLD_GELU, LN_RESID_STATS = 0, 1, 2


def gemm_nt_ln_eligible(N, K):
    """shapes mvptr_gemm_nt_ln takes (256 x 256 tiles of the ping-pong kernel)"""
    return N % 256 == 0 and K % 64 == 0 and K >= 128


def gemm_nt_ln(a, b, mode, bias, stats=None, colsum=None, aux=None, gamma=None, beta=None, out=None):
    """LayerNorm folded into a GEMM (mvptr_gemm_nt_ln).  LN_FOLD_BIAS / LN_FOLD_GELU: a = PRE-LayerNorm rows z [M, K] bf16,
    b = gamma-scaled weight [N, K] bf16, bias = d, stats [M, 2] = (mean, rstd), colsum = c -> bf16 [M, N].
    LN_RESID_STATS: out = a b^T + bias + (aux, or LN(aux) from stats / gamma / beta) -> (out bf16 [M, N], row partials [M, N / 64, 2])."""
    M, K = a.shape
    N = b.shape[0]
    if out is None:
        out = torch.empty((M, N), device=a.device, dtype=torch.bfloat16)
    part = torch.empty((M, N // 64, 2), device=a.device, dtype=torch.float32) if mode == LN_RESID_STATS else None
    _check(load().mvptr_gemm_nt_ln(_p(a), a.stride(0), _p(b), b.stride(0), M, N, K, mode, _p(bias), _p(aux),
                                   aux.stride(0) if aux is not None else 0, _p(stats), _p(colsum), _p(gamma), _p(beta),
                                   _p(out), out.stride(0), _p(part), _stream()))
    return (out, part) if mode == LN_RESID_STATS else out


def ln_stats_finalize(part, H, eps):
    """[M, H / 64, 2] partial sums -> stats [M, 2] = (mean, rstd) (mvptr_ln_stats_finalize)."""
    M = part.shape[0]
    stats = torch.empty((M, 2), device=part.device, dtype=torch.float32)
    _check(load().mvptr_ln_stats_finalize(_p(part), M, H, float(eps), _p(stats), _stream()))
    return stats


def gemm_nt_splitk(a, b, splits, n=None):
    """f32 slabs [splits, M, N] with slab z = A[:, Kz] B[:, Kz]^T over the z-th slice of the reduction index
    (mvptr_gemm_nt_splitk); the caller sums them."""
    M, K = a.shape
    N = b.shape[0] if n is None else n
    slabs = torch.empty((splits, M, N), device=a.device, dtype=torch.float32)
    _check(load().mvptr_gemm_nt_splitk(_p(a), a.stride(0), _p(b), b.stride(0), M, N, K, splits, _p(slabs), N, _stream()))
    return slabs


def gemm_tn(dy, x, dw, n=None, k=None, colsum=None):
    """dw[N,K] += dy[M,N]^T @ x[M,K]  (f32 accumulate into dw); colsum[N] += column sums of dy."""
    assert dy.dtype == torch.bfloat16 and x.dtype == torch.bfloat16 and dw.dtype == torch.float32
    M = dy.shape[0]
    N = dy.shape[1] if n is None else n
    K = x.shape[1] if k is None else k
    assert x.shape[0] == M and dw.stride(1) == 1
    _check(load().mvptr_gemm_tn(_p(dy), dy.stride(0), _p(x), x.stride(0), M, N, K, _p(dw), dw.stride(0), _p(colsum), _stream()))
    return dw


def gemm_tn_multi(problems, slab_workspace=True):
    """problems: list of (dy, x, dw, colsum-or-None); one grouped launch per run of equal M."""
    arr = (TnProblem * len(problems))()
    for q, (dy, x, dw, cs) in zip(arr, problems):
        assert dy.dtype == torch.bfloat16 and x.dtype == torch.bfloat16 and dw.dtype == torch.float32
        assert x.shape[0] == dy.shape[0] and dw.stride(1) == 1 and dw.shape == (dy.shape[1], x.shape[1])
        q.A, q.lda, q.B, q.ldb = dy.data_ptr(), dy.stride(0), x.data_ptr(), x.stride(0)
        q.M, q.N, q.K = dy.shape[0], dy.shape[1], x.shape[1]
        q.dW, q.ldw, q.colsum = dw.data_ptr(), dw.stride(0), (cs.data_ptr() if cs is not None else None)
    lib = load()
    if slab_workspace:
        # few-row launches (6 000 .. 24 000 rows): per-split partial tiles + an ordered reduction instead of f32 atomics
        need = int(lib.mvptr_gemm_tn_ws_bytes(arr, len(problems)))
        if need > 0:
            ws = torch.empty(need, device=problems[0][0].device, dtype=torch.uint8)
            _check(lib.mvptr_gemm_tn_multi_ws(arr, len(problems), _p(ws), need, _stream()))
            return
    _check(lib.mvptr_gemm_tn_multi(arr, len(problems), _stream()))


def gemm_tn_stack(problems, rows_dev=None, max_workgroups=0):
    """problems: list of (dy, x, dw, colsum-or-None) that share M; every weight gradient in ONE balanced launch
    (mvptr_gemm_tn_stack).  rows_dev: device int32 tensor with the rows actually present (<= M), or None."""
    arr = (TnProblem * len(problems))()
    for q, (dy, x, dw, cs) in zip(arr, problems):
        assert dy.dtype == torch.bfloat16 and x.dtype == torch.bfloat16 and dw.dtype == torch.float32
        assert x.shape[0] == dy.shape[0] and dw.stride(1) == 1 and dw.shape == (dy.shape[1], x.shape[1])
        q.A, q.lda, q.B, q.ldb = dy.data_ptr(), dy.stride(0), x.data_ptr(), x.stride(0)
        q.M, q.N, q.K = dy.shape[0], dy.shape[1], x.shape[1]
        q.dW, q.ldw, q.colsum = dw.data_ptr(), dw.stride(0), (cs.data_ptr() if cs is not None else None)
    _check(load().mvptr_gemm_tn_stack(arr, len(problems), _p(rows_dev), int(max_workgroups), _stream()))


def colsum(x, out, n=None):
    _check(load().mvptr_colsum(_p(x), x.stride(0), x.shape[0], x.shape[1] if n is None else n, _p(out), _stream()))
    return out


def attention_fwd(qkv, mask_add, B, L, heads, drop=None, need_lse=True):
    H = heads * 64
    ctx = torch.empty((B * L, H), device=qkv.device, dtype=torch.bfloat16)
    lse = torch.empty((B, heads, L), device=qkv.device, dtype=torch.float32) if need_lse else None
    _check(load().mvptr_attention_fwd(_p(qkv), _p(mask_add), _p(ctx), _p(lse), B, L, heads, _dp(drop), _stream()))
    return ctx, lse


def attention_bwd(qkv, mask_add, ctx, dctx, lse, B, L, heads, drop=None):
    dqkv = torch.empty_like(qkv)
    _check(load().mvptr_attention_bwd(_p(qkv), _p(mask_add), _p(ctx), _p(dctx), _p(lse), _p(dqkv), B, L, heads, _dp(drop), _stream()))
    return dqkv


def attention_probs(qkv, mask_add, B, L, heads):
    """softmax(Q K^T / 8 + mask) as a tensor, f32 [B, heads, L, L] (config.output_attentions; no dropout)."""
    probs = torch.empty((B, heads, L, L), device=qkv.device, dtype=torch.float32)
    _check(load().mvptr_attention_probs(_p(qkv), _p(mask_add), _p(probs), B, L, heads, _stream()))
    return probs


def attention_fwd_packed(qkv, seq_start, seq_len, B, Lmax, heads, mask_add=None, drop=None, need_lse=True):
    """Row-packed sequences: qkv [total_rows, 3H]; sequence b = rows [seq_start[b], +seq_len[b])."""
    H = heads * 64
    ctx = torch.empty((qkv.shape[0], H), device=qkv.device, dtype=torch.bfloat16)
    lse = torch.empty((B, heads, Lmax), device=qkv.device, dtype=torch.float32) if need_lse else None
    _check(load().mvptr_attention_fwd_packed(_p(qkv), _p(mask_add), _p(ctx), _p(lse), _p(seq_start), _p(seq_len), B, Lmax, heads,
                                             _dp(drop), _stream()))
    return ctx, lse


def attention_bwd_packed(qkv, seq_start, seq_len, ctx, dctx, lse, B, Lmax, heads, mask_add=None, drop=None):
    dqkv = torch.empty_like(qkv)
    _check(load().mvptr_attention_bwd_packed(_p(qkv), _p(mask_add), _p(ctx), _p(dctx), _p(lse), _p(dqkv), _p(seq_start), _p(seq_len),
                                             B, Lmax, heads, _dp(drop), _stream()))
    return dqkv


def layernorm_fwd(z, gamma, beta, eps, out=None, rows_per_group=None, group_stride=0, row_offset=0,
                  drop=None, save_stats=True):
    M, H = z.shape
    if out is None:
        out = torch.empty_like(z)
    mean = torch.empty(M, device=z.device, dtype=torch.float32) if save_stats else None
    rstd = torch.empty(M, device=z.device, dtype=torch.float32) if save_stats else None
    _check(load().mvptr_layernorm_fwd(_p(z), _p(gamma), _p(beta), float(eps), _p(out), _p(mean), _p(rstd),
                                      M, H, rows_per_group or M, group_stride, row_offset, _dp(drop), _stream()))
    return out, mean, rstd


def layernorm_bwd(dy, z, mean, rstd, gamma, dgamma, dbeta, dbias=None, rows_per_group=None,
                  group_stride=0, row_offset=0, y_drop=None, dense_drop=None):
    M, H = z.shape
    dz = torch.empty_like(z)
    dd = torch.empty_like(z) if dense_drop is not None else None
    nws = load().mvptr_layernorm_bwd_ws_bytes(M, H)
    ws = torch.empty(nws, device=z.device, dtype=torch.uint8)
    _check(load().mvptr_layernorm_bwd(_p(dy), _p(z), _p(mean), _p(rstd), _p(gamma), _p(dz), _p(dd),
                                      _p(dgamma), _p(dbeta), _p(dbias), M, H, rows_per_group or M,
                                      group_stride, row_offset, _dp(y_drop), _dp(dense_drop), _p(ws), nws, _stream()))
    return dz, dd


def embed_fwd(ids, pos_ids, type_ids, word, pos, typ):
    rows = ids.numel()
    H = word.shape[1]
    z = torch.empty((rows, H), device=word.device, dtype=torch.bfloat16)
    _check(load().mvptr_embed_fwd(_p(ids), _p(pos_ids), _p(type_ids), _p(word), _p(pos), _p(typ), _p(z),
                                  rows, H, word.shape[0], pos.shape[0], typ.shape[0], _stream()))
    return z


def embed_bwd(ids, pos_ids, type_ids, dz, dword, dpos, dtype_):
    rows, H = dz.shape
    _check(load().mvptr_embed_bwd(_p(ids), _p(pos_ids), _p(type_ids), _p(dz), _p(dword), _p(dpos), _p(dtype_), rows, H, _stream()))


CAST_TASK_DTYPE = [("src", "<u8"), ("ld_src", "<i8"), ("rows", "<i4"), ("cols", "<i4"), ("dst", "<u8"), ("ld_dst", "<i8"),
                   ("dst_t", "<u8"), ("ld_dst_t", "<i8"), ("col_off_t", "<i4"), ("tiles_x", "<i4"), ("dst_f32", "<u8")]


class CastPlan:
    """Device-resident task table for mvptr_cast_multi.  add() jobs, build() once, run() per refresh."""

    def __init__(self, device):
        self.device, self.jobs, self.table, self.base, self.total = device, [], None, None, 0

    def add(self, src, dst=None, dst_t=None, col_off_t=0, dst_f32=None):
        assert src.dtype == torch.float32 and src.is_contiguous()
        src2 = src if src.dim() == 2 else src.view(1, -1)
        rows, cols = src2.shape
        ld_dst = dst.stride(0) if dst is not None else cols
        self.jobs.append((src2, rows, cols, dst, ld_dst, dst_t, col_off_t, dst_f32))

    def build(self):
        import numpy as np
        tab = np.zeros(len(self.jobs), dtype=CAST_TASK_DTYPE)
        base = np.zeros(len(self.jobs) + 1, dtype=np.int32)
        for i, (src, rows, cols, dst, ld_dst, dst_t, col_off_t, dst_f32) in enumerate(self.jobs):
            tx = (max(cols, ld_dst if dst is not None else cols) + 31) // 32
            tab[i] = (src.data_ptr(), src.stride(0), rows, cols, dst.data_ptr() if dst is not None else 0, ld_dst,
                      dst_t.data_ptr() if dst_t is not None else 0, dst_t.stride(0) if dst_t is not None else 0,
                      col_off_t, tx, dst_f32.data_ptr() if dst_f32 is not None else 0)
            base[i + 1] = base[i] + tx * ((rows + 31) // 32)
        self.table = torch.from_numpy(tab.view(np.uint8).copy()).to(self.device)
        self.base = torch.from_numpy(base).to(self.device)
        self.total = int(base[-1])

    def run(self):
        _check(load().mvptr_cast_multi(_p(self.table), _p(self.base), len(self.jobs), self.total, _stream()))


def cast_pack(src, dst=None, dst_t=None, col_off_t=0):
    """f32 [rows, cols] -> bf16 dst [rows, ld>=cols] (zero padded) and/or transposed dst_t."""
    assert src.dtype == torch.float32 and src.dim() == 2 and src.stride(1) == 1
    rows, cols = src.shape
    _check(load().mvptr_cast_pack(_p(src), src.stride(0), rows, cols, _p(dst), dst.stride(0) if dst is not None else 0,
                                  _p(dst_t), dst_t.stride(0) if dst_t is not None else 0, col_off_t, _stream()))


def decoder_ce_fwd(h, w, bias, labels, V):
    """Fused decoder GEMM + cross entropy (no logits tensor): returns (loss_row f32 [M], lse_row f32 [M])."""
    assert h.dtype == torch.bfloat16 and w.dtype == torch.bfloat16 and labels.dtype == torch.int64
    M, K = h.shape
    part = torch.empty((M, (V + 63) // 64, 2), device=h.device, dtype=torch.float32)
    lab_logit = torch.zeros(M, device=h.device, dtype=torch.float32)
    loss_row = torch.empty(M, device=h.device, dtype=torch.float32)
    lse_row = torch.empty(M, device=h.device, dtype=torch.float32)
    _check(load().mvptr_decoder_ce_fwd(_p(h), h.stride(0), _p(w), w.stride(0), _p(bias), _p(labels), M, V, K,
                                       _p(part), _p(lab_logit), _p(loss_row), _p(lse_row), _stream()))
    return loss_row, lse_row


def decoder_ce_bwd(h, w, bias, labels, lse_row, scale, V, Vp):
    """d = (softmax(h w^T + bias) - onehot(labels)) * scale as bf16 [M, Vp], logits recomputed in the GEMM."""
    M, K = h.shape
    d = torch.empty((M, Vp), device=h.device, dtype=torch.bfloat16)
    _check(load().mvptr_decoder_ce_bwd(_p(h), h.stride(0), _p(w), w.stride(0), _p(bias), _p(labels), _p(lse_row),
                                       _p(scale), M, V, K, _p(d), d.stride(0), Vp, _stream()))
    return d


def _diag_lib():
    lib = load()
    if not hasattr(lib, "mvptr_diag_stream_read"):
        raise RuntimeError("the measurement helpers live in the diagnostic library only: `make -C mvp_pytorch_amd/csrc diag` and MVPTR_LIB=diag")
    return lib


def diag_store_probe(buf, blocks, bytes_per_wave, rows_per_instr, stride):
    """Measurement helper: store-rate probe (see mvptr_diag.h)."""
    _check(_diag_lib().mvptr_diag_store_probe(_p(buf), buf.numel() * buf.element_size(), blocks, bytes_per_wave, rows_per_instr,
                                         stride, _stream()))


def diag_fill_probe(buf, blocks, wg_bytes, reps, shared, mode, sink):
    """Measurement helper: operand-fill probe (see mvptr_diag.h)."""
    _check(_diag_lib().mvptr_diag_fill_probe(_p(buf), buf.numel() * buf.element_size(), blocks, wg_bytes, reps, int(shared), mode,
                                        _p(sink), _stream()))


def diag_stream_read(buf, mode, sink):
    """Calibration helper: read every byte of `buf` once (mode 0: LDS-DMA, mode 1: global loads)."""
    nbytes = buf.numel() * buf.element_size()
    _check(_diag_lib().mvptr_diag_stream_read(_p(buf), nbytes, mode, _p(sink), _stream()))


def b64_decode_features(text, offsets, n_chars, num_boxes, R, D, out_f32=None, out_bf16=None, err=None):
    """Region features of a batch, still the base64 text of the TSV rows (uint8 device tensor,
    samples at 16-byte aligned `offsets`), -> f32 [n, R, D] and / or K-padded bf16 [n*R, ld]
    (oscar_tsv4.py:696-724 + the padding of :332-352).  err: int32[1] device flag (see mvptr.h)."""
    n = offsets.shape[0]
    assert text.dtype == torch.uint8 and offsets.dtype == torch.int64 and n_chars.dtype == torch.int64
    assert num_boxes.dtype == torch.int32 and num_boxes.shape[0] == n and n_chars.shape[0] == n
    if out_f32 is not None:
        assert out_f32.dtype == torch.float32 and out_f32.is_contiguous() and out_f32.numel() == n * R * D
    ld = 0
    if out_bf16 is not None:
        assert out_bf16.dtype == torch.bfloat16 and out_bf16.stride(1) == 1 and out_bf16.shape[0] == n * R
        ld = out_bf16.stride(0)
    if err is None:
        err = torch.zeros(1, device=text.device, dtype=torch.int32)
    _check(load().mvptr_b64_decode_features(_p(text), _p(offsets), _p(n_chars), _p(num_boxes), n, R, D, _p(out_f32),
                                            _p(out_bf16), ld, _p(err), _stream()))
    return err


def cast_f32(src, rows=None, cols=None):
    rows = src.shape[0] if rows is None else rows
    cols = src.shape[1] if cols is None else cols
    dst = torch.empty((rows, cols), device=src.device, dtype=torch.float32)
    _check(load().mvptr_cast_f32(_p(src), src.stride(0), rows, cols, _p(dst), cols, _stream()))
    return dst


def ce_fwd(logits, labels, V=None):
    M = logits.shape[0]
    V = logits.shape[1] if V is None else V
    loss = torch.empty(M, device=logits.device, dtype=torch.float32)
    lse = torch.empty(M, device=logits.device, dtype=torch.float32)
    _check(load().mvptr_ce_fwd(_p(logits), logits.stride(0), _p(labels), _p(loss), _p(lse), M, V, _stream()))
    return loss, lse


def ce_bwd(logits, labels, lse, scale, V, Vpad):
    M = logits.shape[0]
    d = torch.empty((M, Vpad), device=logits.device, dtype=torch.bfloat16)
    _check(load().mvptr_ce_bwd(_p(logits), logits.stride(0), _p(labels), _p(lse), _p(scale), _p(d), Vpad, M, V, Vpad, _stream()))
    return d


def dropout_mask(drop, n, device):
    keep = torch.empty(n, device=device, dtype=torch.uint8)
    _check(load().mvptr_dropout_mask(_dp(drop), n, _p(keep), _stream()))
    return keep


ADAMW_CHUNK = 8192      # elements per workgroup: 32 per thread; 65536 left the 28 M non-mirrored parameters on 430 workgroups (1.8 TB/s)


def adamw_multi(table_dev, chunk_tensor_dev, chunk_offset_dev, n_chunks, beta1, beta2, eps, grad_scale=None):
    """Fused multi-tensor AdamW (see mvptr_adamw_multi); tables are device tensors built by
    mvp_pytorch_amd.optimization.AdamW.  grad_scale: device f32 scalar (clip coefficient) or None."""
    _check(load().mvptr_adamw_multi(_p(table_dev), _p(chunk_tensor_dev), _p(chunk_offset_dev), n_chunks,
                                    ADAMW_CHUNK, float(beta1), float(beta2), float(eps), _p(grad_scale), _stream()))


MIRROR_DT = [("p", "<u8"), ("g", "<u8"), ("m", "<u8"), ("v", "<u8"), ("rows", "<i4"), ("cols", "<i4"), ("step_size", "<f4"),
             ("decay", "<f4"), ("dst", "<u8"), ("ld_dst", "<i8"), ("dst_t", "<u8"), ("ld_dst_t", "<i8"), ("col_off_t", "<i4"),
             ("pad_", "<i4"), ("dst_f32", "<u8")]          # mvptr_adamw_mirror_tensor


def adamw_mirror_multi(table_dev, tile_base_dev, n_tensors, total_tiles, beta1, beta2, eps, grad_scale=None):
    """AdamW + bf16 working copies in one pass over 64 x 64 tiles (see mvptr_adamw_mirror_multi)."""
    _check(load().mvptr_adamw_mirror_multi(_p(table_dev), _p(tile_base_dev), n_tensors, total_tiles, float(beta1), float(beta2),
                                           float(eps), _p(grad_scale), _stream()))


def grad_clip_coef(flats, max_norm, scratch=None):
    """Global 2-norm of the flat f32 gradient buffers and the clip coefficient min(1, max_norm / (norm + 1e-6)) as
    device scalars -> (norm [1], coef [1], scratch).  Two passes: per-16K-element partial sums of squares, then one
    workgroup that adds them in index order (mvptr_sumsq_partial / mvptr_clip_coef)."""
    lib = load()
    counts = [int(lib.mvptr_sumsq_partials(f.numel())) for f in flats]
    total = sum(counts)
    dev = flats[0].device
    if scratch is None or scratch.numel() < total + 2:
        scratch = torch.empty(total + 2, device=dev, dtype=torch.float32)
    o = 0
    for f, c in zip(flats, counts):
        assert f.dtype == torch.float32 and f.is_contiguous()
        _check(lib.mvptr_sumsq_partial(_p(f), f.numel(), c_void_p(scratch.data_ptr() + 4 * o), _stream()))
        o += c
    norm, coef = scratch[total:total + 1], scratch[total + 1:total + 2]
    _check(lib.mvptr_clip_coef(_p(scratch), total, float(max_norm), _p(norm), _p(coef), _stream()))
    return norm, coef, scratch


def sgemm_small(a, b, trans_a=False, trans_b=False, a_rows=None, b_rows=None, bias=None, act=None, alpha=1.0, out=None,
                accumulate=False, m=None, n=None, k=None):
    """C = act(alpha * op(A) op(B) + bias) in f32 (mvptr_sgemm_small).  a / b: 2-D f32 or bf16 tensors with unit
    column stride; a_rows: int32 gather of A's stored rows; out: f32 [M, N] (accumulated into when accumulate)."""
    assert a.stride(-1) == 1 and b.stride(-1) == 1
    if trans_a:
        K_, M_ = (a_rows.numel() if a_rows is not None else a.shape[0]), a.shape[1]
    else:
        M_, K_ = (a_rows.numel() if a_rows is not None else a.shape[0]), a.shape[1]
    N_ = (b_rows.numel() if b_rows is not None else b.shape[0]) if trans_b else b.shape[1]
    M_, N_, K_ = m or M_, n or N_, k or K_
    if out is None:
        out = torch.empty((M_, N_), device=a.device, dtype=torch.float32)
    _check(load().mvptr_sgemm_small(_p(a), a.stride(0), int(a.dtype == torch.bfloat16), int(trans_a), _p(a_rows), _p(b), b.stride(0),
                                    int(b.dtype == torch.bfloat16), int(trans_b), _p(b_rows), M_, N_, K_, float(alpha), _p(bias),
                                    1 if act == "tanh" else 0, int(accumulate), _p(out), out.stride(0), _stream()))
    return out


def l2norm_fwd(y, eps=1e-12):
    g, inv = torch.empty_like(y), torch.empty(y.shape[0], device=y.device, dtype=torch.float32)
    _check(load().mvptr_l2norm_fwd(_p(y), _p(g), _p(inv), y.shape[0], y.shape[1], float(eps), _stream()))
    return g, inv


def l2norm_bwd(g, inv, dg):
    dy = torch.empty_like(g)
    _check(load().mvptr_l2norm_bwd(_p(g), _p(inv), _p(dg), _p(dy), g.shape[0], g.shape[1], _stream()))
    return dy


def clip_ce_fwd(sim, logit_scale):
    n = sim.shape[0]
    ws = torch.empty(4 * n + 1, device=sim.device, dtype=torch.float32)
    lse, parts, loss = ws[:2 * n], ws[2 * n:4 * n], ws[4 * n:]
    _check(load().mvptr_clip_ce_fwd(_p(sim), n, sim.stride(0), _p(logit_scale), _p(lse), _p(parts), _p(loss), _stream()))
    return loss.reshape(()), lse


def clip_ce_bwd(sim, logit_scale, lse, gloss, dlogit_scale=None):
    n = sim.shape[0]
    dsim = torch.empty((n, n), device=sim.device, dtype=torch.float32)
    parts = torch.empty(n, device=sim.device, dtype=torch.float32)
    _check(load().mvptr_clip_ce_bwd(_p(sim), n, sim.stride(0), _p(logit_scale), _p(lse), _p(gloss), _p(dsim), _p(parts),
                                    _p(dlogit_scale), _stream()))
    return dsim


def ce_mean_small(logits, labels, want_grad=True):
    """-> (mean CE over the rows with a label in [0, V), d loss / d logits f32 [M, V] or None)."""
    M, V = logits.shape
    loss = torch.empty(1, device=logits.device, dtype=torch.float32)
    d = torch.empty((M, V), device=logits.device, dtype=torch.float32) if want_grad else None
    _check(load().mvptr_ce_mean_small(_p(logits), logits.stride(0), _p(labels), M, V, _p(loss), _p(d), _stream()))
    return loss.reshape(()), d


def hard_negative_mine(sim, perm=None, want_sel=False):
    """mvptr_hard_negative_mine: sim f32 [n, n] -> (hard_img, hard_txt) int64 [n]; with perm (torch.randperm(n) on the
    device) also (hard_txt_full, hard_img_full) and, with want_sel, the [2n] `sel` vectors of the joint pack maps."""
    n = sim.shape[0]
    assert sim.dtype == torch.float32 and sim.dim() == 2 and sim.shape[1] == n and sim.stride(1) == 1
    i64 = dict(device=sim.device, dtype=torch.int64)
    hard_img, hard_txt = torch.empty(n, **i64), torch.empty(n, **i64)
    if perm is None:
        _check(load().mvptr_hard_negative_mine(_p(sim), n, sim.stride(0), None, _p(hard_img), _p(hard_txt), None, None, None, None, _stream()))
        return hard_img, hard_txt
    perm = perm.to(torch.int64).contiguous()
    htf, hif = torch.empty(n, **i64), torch.empty(n, **i64)
    st, si = (torch.empty(2 * n, **i64), torch.empty(2 * n, **i64)) if want_sel else (None, None)
    _check(load().mvptr_hard_negative_mine(_p(sim), n, sim.stride(0), _p(perm), _p(hard_img), _p(hard_txt), _p(htf), _p(hif), _p(st), _p(si),
                                           _stream()))
    return hard_img, hard_txt, htf, hif, st, si


def bce_logits(logits, labels, want_grad=True):
    """mvptr_bce_logits: instance_bce_with_logits (vl:878-883) -> (loss f32 [], d loss / d logits or None)."""
    rows, cols = logits.shape
    assert logits.dtype == torch.float32 and labels.dtype == torch.float32 and logits.is_contiguous() and labels.is_contiguous()
    n_parts = max(1, min(1024, (rows * cols + 8191) // 8192))
    loss = torch.empty(1, device=logits.device, dtype=torch.float32)
    parts = torch.empty(n_parts, device=logits.device, dtype=torch.float32)
    d = torch.empty_like(logits) if want_grad else None
    _check(load().mvptr_bce_logits(_p(logits), _p(labels), rows, cols, _p(loss), _p(d), _p(parts), n_parts, _stream()))
    return loss.reshape(()), d


def gather_rows(src, idx, out=None, src2=None):
    """out[i] = src[idx[i]] (bf16 rows; idx int32, negative = zero row); with src2, idx >= src.shape[0] reads
    src2[idx - src.shape[0]]."""
    n, H = idx.numel(), src.shape[1]
    if out is None:
        out = torch.empty((n, H), device=src.device, dtype=torch.bfloat16)
    if n:
        _check(load().mvptr_gather_rows(_p(src), src.stride(0), _p(src2), src2.stride(0) if src2 is not None else 0,
                                        src.shape[0], _p(idx), _p(out), out.stride(0), n, H, _stream()))
    return out


def scatter_add_rows(src, idx, dst, dst2=None):
    """dst[idx[i]] += src[i] (dst f32 or bf16 rows, src bf16 or f32; atomics, rows may repeat); with dst2,
    idx >= dst.shape[0] adds into dst2[idx - dst.shape[0]]."""
    n, H = idx.numel(), dst.shape[1]
    if n:
        _check(load().mvptr_scatter_add_rows(_p(src), src.stride(0), int(src.dtype == torch.float32), _p(idx), _p(dst), dst.stride(0),
                                             _p(dst2), dst2.stride(0) if dst2 is not None else 0, dst.shape[0],
                                             int(dst.dtype == torch.float32), n, H, _stream()))
    return dst


# ---------------------------------------------------------------------------------------------
# Device error word (ABI 7): data-dependent failures that the reference raises catchably (modeling_vlbert.py:435,542,1547) are
# detected by kernels / queued device ops long after the host has moved on.  Rounds 4-5 trapped or used torch._assert_async — on
# ROCm both abort the whole process.  Now such a check writes a code into a small per-device int64 word; the host looks at it
# whenever it reads ANYTHING back from that device (engine.AsyncCounts carries word[0] along with every count copy;
# check_device_errors() reads it explicitly) and raises RuntimeError there, one step late at most, with the process intact.
DEV_ERR_SCORED_ROWS, DEV_ERR_PHRASES, DEV_ERR_FEW_REGIONS = 1, 2, 3
DEV_ERR_TEXT = {
    DEV_ERR_SCORED_ROWS: "more scored (label > -1) slots than the host's count: host_counts.scored_* does not describe this batch",
    DEV_ERR_PHRASES: "a sample has more phrases than config.max_phrases",
    DEV_ERR_FEW_REGIONS: "word-region alignment needs >= 3 valid regions per image (topk(3), modeling_vlbert.py:1547)",
}
_err_words = {}


def device_error_word(device):
    """int64 [4] on `device` (code, detail, detail, unused); zero = no error"""
    key = torch.device(device).index if torch.device(device).type == "cuda" else -1
    if key == -1:
        return None
    w = _err_words.get(key)
    if w is None:
        w = _err_words[key] = torch.zeros(4, dtype=torch.int64, device=device)
    return w


def flag_device_error_if(bad, code):
    """queue `if bad: word[0] = code` on bad's device (bad: 0-dim bool tensor, e.g. `(x > limit).any()`); ONE small launch, no
    host sync (torch._assert_async cost the same launch and aborted the process)"""
    w = device_error_word(bad.device)
    if w is None:       # CPU tensors: the host can simply look
        if bool(bad):
            raise RuntimeError(DEV_ERR_TEXT[code])
        return
    w[:1].masked_fill_(bad.reshape(1), code)


def raise_device_error(code, device=None):
    """host side of the error word: called with word[0] as read back; clears the word and raises"""
    code = int(code)
    if code == 0:
        return
    detail = ""
    if device is not None:
        w = device_error_word(device)
        if w is not None:
            vals = [int(v) for v in w.tolist()]
            if code == DEV_ERR_SCORED_ROWS:
                detail = " (%d scored rows, %d slots)" % (vals[1], vals[2])
            w.zero_()
    raise RuntimeError("device-side check failed: " + DEV_ERR_TEXT.get(code, "error code %d" % code) + detail)


def check_device_errors(device=None):
    """read the error word(s) now (synchronises) and raise if a device-side check has failed since the last look"""
    for key, w in list(_err_words.items()):
        if device is not None and torch.device(device).index != key:
            continue
        raise_device_error(int(w[0].item()), w.device)


_salt_words = {}


def dropout_salt(device):
    """The dropout salt word of `device` (int32 [1], zero; registered with the library on first use — mvptr_set_dropout_salt).
    Kernels mix it into their dropout seeds; a graph-captured step adds 1 per replay (train.GraphedStep), eager steps leave it 0."""
    d = torch.device(device)
    key = d.index if d.index is not None else torch.cuda.current_device()
    w = _salt_words.get(key)
    if w is None:
        w = _salt_words[key] = torch.zeros(1, dtype=torch.int32, device=d)
        with torch.cuda.device(key):
            _check(load().mvptr_set_dropout_salt(_p(w)))
    return w


def compact_scored(labels, pos, n_out):
    """labels int64 [B, L] (contiguous), pos int32 [*, ld] row map (its first B rows / L columns are read; None: flat slot index)
    -> (labels of the slots with label > -1 in ascending slot order int64 [n_out], their rows int32 [n_out]) — mvptr_compact_scored."""
    B, L = labels.shape
    assert labels.dtype == torch.int64 and labels.is_contiguous()
    assert pos is None or (pos.dtype == torch.int32 and pos.stride(-1) == 1 and pos.shape[0] >= B and pos.shape[1] >= L)
    ol = torch.empty(n_out, dtype=torch.int64, device=labels.device)
    orow = torch.empty(n_out, dtype=torch.int32, device=labels.device)
    _check(load().mvptr_compact_scored(_p(labels), _p(pos), pos.stride(0) if pos is not None else 0, B, L, int(n_out), _p(ol), _p(orow),
                                       _p(device_error_word(labels.device)), _stream()))
    return ol, orow


def masked_mean(loss_row, labels):
    """(mean of loss_row over the rows with label >= 0 (0-dim f32), max(count, 1) (0-dim f32)) — mvptr_masked_mean."""
    out = torch.empty(2, dtype=torch.float32, device=loss_row.device)
    _check(load().mvptr_masked_mean(_p(loss_row), _p(labels), loss_row.numel(), _p(out), _stream()))
    return out[0], out[1]


def dgelu_mul(dy, stash, Npad=None):
    """bf16 [M, Npad] = dy * gelu'(u) decoded from the stash (uint8: 8-bit fixed point; bfloat16: the bf16 stash; columns
    N .. Npad zero) — mvptr_dgelu_mul."""
    M, N = dy.shape
    Npad = N if Npad is None else Npad
    assert stash.dtype in (torch.uint8, torch.bfloat16)
    out = torch.empty((M, Npad), dtype=torch.bfloat16, device=dy.device)
    _check(load().mvptr_dgelu_mul(_p(dy), dy.stride(0), _p(stash), stash.stride(0), 1 if stash.dtype == torch.bfloat16 else 0, _p(out),
                                  out.stride(0), M, N, Npad, _stream()))
    return out


def tap_rows_bwd(taps, rows, rows2, H):
    """Gradient of a multi-tap row gather: taps = [(g [n, H] bf16 / f32 contiguous rows, idx int32 [n])] -> (d bf16 [rows, H],
    d2 bf16 [rows2, H] or None); d[r] = sum of the g rows whose idx is r (f32 sums, one rounding; untapped rows are zero rows;
    idx >= rows addresses d2) — mvptr_tap_rows_bwd."""
    assert 1 <= len(taps) <= TAP_MAX
    dev = taps[0][0].device
    arr = (Tap * len(taps))()
    total = 0
    keep = []
    for k, (g, idx) in enumerate(taps):
        assert g.dim() == 2 and g.shape[1] == H and g.stride(1) == 1 and g.dtype in (torch.bfloat16, torch.float32)
        assert idx.dtype == torch.int32 and idx.numel() == g.shape[0]
        idx = idx.contiguous()
        keep.append(idx)
        arr[k].g, arr[k].ld_g, arr[k].idx, arr[k].n, arr[k].g_f32 = g.data_ptr(), g.stride(0), idx.data_ptr(), g.shape[0], int(g.dtype == torch.float32)
        total += g.shape[0]
    d = torch.empty(rows, H, dtype=torch.bfloat16, device=dev)
    d2 = torch.empty(rows2, H, dtype=torch.bfloat16, device=dev) if rows2 else None
    work = torch.empty(2 * (rows + rows2 + 1) + total, dtype=torch.int32, device=dev)
    _check(load().mvptr_tap_rows_bwd(arr, len(taps), _p(d), d.stride(0), rows, _p(d2), d2.stride(0) if d2 is not None else 0, rows2, H,
                                     _p(work), work.numel(), _stream()))
    return d, d2


def pack_maps(segs, n_seq, fill_idx=False):
    """segs: 1 or 2 dicts(mask f32 [rows, ld] additive, sel int64 [n_seq] or None, col0, len, pos int32 [rows, ld_pos] or
    None, src_seq_stride, src_base) -> (pos int32 [n_seq, Ltot], idx int32 [n_seq * Ltot] (first `rows` entries valid),
    seq_start int32 [n_seq], seq_len int32 [n_seq], counts int64 [2] = rows, longest) — see mvptr_pack_maps."""
    dev = segs[0]["mask"].device
    arr = (PackSeg * len(segs))()
    Ltot = 0
    for k, sg in enumerate(segs):
        m = sg["mask"]
        assert m.dtype == torch.float32 and m.stride(-1) == 1
        a = arr[k]
        a.mask, a.ld_mask = m.data_ptr(), m.stride(0)
        a.sel = sg["sel"].data_ptr() if sg.get("sel") is not None else None
        a.col0, a.len = int(sg.get("col0", 0)), int(sg["len"])
        pos = sg.get("pos")
        a.pos, a.ld_pos = (pos.data_ptr(), pos.stride(0)) if pos is not None else (None, 0)
        a.src_seq_stride, a.src_base = int(sg.get("src_seq_stride", 0)), int(sg.get("src_base", 0))
        Ltot += a.len
    pos_out = torch.empty((n_seq, Ltot), device=dev, dtype=torch.int32)
    # fill_idx: entries past the valid rows read -1 (a gather of the WHOLE vector then yields zero rows there: the
    # sync-free joint pass, whose row count stays on the device)
    idx_out = (torch.full if fill_idx else torch.empty)(*(((n_seq * Ltot,), -1) if fill_idx else ((n_seq * Ltot,),)), device=dev, dtype=torch.int32)
    seq_start = torch.empty(n_seq, device=dev, dtype=torch.int32)
    seq_len = torch.empty(n_seq, device=dev, dtype=torch.int32)
    counts = torch.empty(2, device=dev, dtype=torch.int64)
    _check(load().mvptr_pack_maps(arr, len(segs), n_seq, _p(pos_out), _p(idx_out), _p(seq_start), _p(seq_len), _p(counts), _stream()))
    return pos_out, idx_out, seq_start, seq_len, counts


def check_counts(counts_a, counts_b, expect):
    """mvptr_check_counts: the device-side counts of two pack_maps calls (int64 [2] each: rows, longest) must equal the
    host's numbers `expect` = (rows_a, lmax_a, rows_b, lmax_b); a mismatch traps the kernel (the process aborts)."""
    _check(load().mvptr_check_counts(_p(counts_a), _p(counts_b), int(expect[0]), int(expect[1]), int(expect[2]), int(expect[3]), _stream()))


def wra_rows(pos, phrase_index, img_index, n, Pw, Rw):
    """pos int32 [>= n, Lj] (packed row of every slot, hip.pack_maps) -> rows_p int32 [n, Pw], rows_r int32 [n, Rw]:
    the packed rows of each sample's phrase / region slots, -1 beyond its counts (mvptr_wra_rows)."""
    assert pos.dtype == torch.int32 and pos.is_contiguous() and pos.shape[0] >= n
    phrase_index, img_index = phrase_index.contiguous(), img_index.contiguous()
    assert phrase_index.dtype == torch.int64 and img_index.dtype == torch.int64
    rows_p = torch.empty((n, Pw), device=pos.device, dtype=torch.int32)
    rows_r = torch.empty((n, Rw), device=pos.device, dtype=torch.int32)
    _check(load().mvptr_wra_rows(_p(pos), pos.shape[1], _p(phrase_index), _p(img_index), n, Pw, Rw, _p(rows_p), _p(rows_r), _stream()))
    return rows_p, rows_r


def wra_fwd(txt, reg, phrase_index, img_index, pos_pick, neg_pick, neg_img):
    """txt bf16 [n, Pw, H], reg bf16 [n, Rw, H], draws int64 -> (loss f32 [1], saved tuple for wra_bwd) — mvptr_wra_fwd."""
    n, Pw, H = txt.shape
    Rw = reg.shape[1]
    dev = txt.device
    for t in (txt, reg, phrase_index, img_index, pos_pick, neg_pick, neg_img):
        assert t.is_contiguous()
    assert txt.dtype == torch.bfloat16 and reg.dtype == torch.bfloat16 and reg.shape[0] == n and reg.shape[2] == H
    assert all(t.dtype == torch.int64 for t in (phrase_index, img_index, pos_pick, neg_pick, neg_img))
    assert pos_pick.shape == (n, Pw) and neg_pick.shape == (n, Pw) and neg_img.shape == (n,)
    f32 = torch.empty(1 + 2 * n + 3 * n * Pw + n * Rw, device=dev, dtype=torch.float32)
    i32 = torch.empty(2 * n + 2 * n * Pw, device=dev, dtype=torch.int32)
    loss, hinge, coef, sval, inv_p, inv_r = f32.split([1, n, n, 2 * n * Pw, n * Pw, n * Rw])
    cnt, sel = i32.split([2 * n, 2 * n * Pw])
    _check(load().mvptr_wra_fwd(_p(txt), _p(reg), _p(phrase_index), _p(img_index), _p(pos_pick), _p(neg_pick), _p(neg_img),
                                n, Pw, Rw, H, _p(loss), _p(hinge), _p(coef), _p(cnt), _p(sel), _p(sval), _p(inv_p), _p(inv_r),
                                _stream()))
    return loss, (cnt, sel, sval, inv_p, inv_r, coef, hinge)


def wra_bwd(txt, reg, neg_img, saved, gout):
    """-> d_txt bf16 like txt, d_reg bf16 like reg; gout: f32 tensor with one element on the device (mvptr_wra_bwd)."""
    n, Pw, H = txt.shape
    Rw = reg.shape[1]
    cnt, sel, sval, inv_p, inv_r, coef, _ = saved
    gout = gout.reshape(1).to(torch.float32)
    d_txt, d_reg = torch.empty_like(txt), torch.empty_like(reg)
    _check(load().mvptr_wra_bwd(_p(txt), _p(reg), _p(neg_img), n, Pw, Rw, H, _p(cnt), _p(sel), _p(sval), _p(inv_p), _p(inv_r),
                                _p(coef), _p(gout), _p(d_txt), _p(d_reg), _stream()))
    return d_txt, d_reg
