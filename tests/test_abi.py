"""The C-ABI library loads and exports every symbol include/mvptr.h declares (no compute)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_symbols():
    text = open(os.path.join(ROOT, "include", "mvptr.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(mvptr_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_header_symbols():
    from mvp_pytorch_amd import hip
    if not os.path.exists(hip.LIB_PATH):
        import __graft_entry__ as g
        g.build()
    lib = ctypes.CDLL(hip.LIB_PATH)
    syms = _header_symbols()
    assert len(syms) >= 18
    for s in syms:
        assert hasattr(lib, s), "missing export: " + s
    assert sorted(hip.SYMBOLS) == syms
    # the measurement helpers are declared in mvptr_diag.h and live in the diagnostic build only
    for s in hip.DIAG_SYMBOLS:
        assert not hasattr(lib, s), "diagnostic export in the product library: " + s
    dtext = open(os.path.join(ROOT, "include", "mvptr_diag.h")).read()
    dtext = re.sub(r"/\*.*?\*/", "", dtext, flags=re.S)
    assert sorted(set(re.findall(r"\b(mvptr_[a-z0-9_]+)\s*\(", dtext))) == sorted(hip.DIAG_SYMBOLS)


def test_abi_version_and_error_string():
    from mvp_pytorch_amd import hip
    lib = hip.load()
    assert hip.query(0) == 7   # MVPTR_ABI_VERSION (7: device error word of mvptr_compact_scored, mvptr_diag_* out of the product library; 6: mvptr_gemm_nt_ln + mvptr_ln_stats_finalize; 5: mvptr_gemm_tn_stack + mvptr_encoder_layer_bwd_defer; 4: 8-bit gelu' stash)
    # argument validation happens on the host before any launch: safe without a GPU
    rc = lib.mvptr_gemm_nt(None, 8, None, 8, 0, 8, 8, 0, None, None, 0, None, None, 8, None, None, None)
    assert rc == -1
    assert b"gemm_nt" in lib.mvptr_last_error()
    d = hip.LayerDesc(2, 300, 128, 2, 512, 1e-12, 1, 0, 0, 0)
    assert lib.mvptr_layer_saved_bytes(ctypes.byref(d)) == -1  # L > 256
    d = hip.LayerDesc(2, 40, 128, 2, 512, 1e-12, 1, 0, 0, 0)
    dense = lib.mvptr_layer_saved_bytes(ctypes.byref(d))
    assert dense > 0
    assert lib.mvptr_layer_workspace_bytes(ctypes.byref(d)) > 0
    # row-packed mode: M rows + both sequence arrays, or none of them
    d = hip.LayerDesc(2, 40, 128, 2, 512, 1e-12, 1, 0, 0, 0, 50, 0, None, None)
    assert lib.mvptr_layer_saved_bytes(ctypes.byref(d)) == -1
    assert b"seq_start" in lib.mvptr_last_error()
    d = hip.LayerDesc(2, 40, 128, 2, 512, 1e-12, 1, 0, 0, 0, 81, 0, 4096, 4096)
    assert lib.mvptr_layer_saved_bytes(ctypes.byref(d)) == -1  # M > B*L
    d = hip.LayerDesc(2, 40, 128, 2, 512, 1e-12, 1, 0, 0, 0, 50, 0, 4096, 4096)
    assert 0 < lib.mvptr_layer_saved_bytes(ctypes.byref(d)) < dense
    # device-side row count: only in row-packed mode, planning hint inside [0, M]
    d = hip.LayerDesc(2, 40, 128, 2, 512, 1e-12, 1, 0, 0, 0, 0, 0, None, None, 4096)
    assert lib.mvptr_layer_saved_bytes(ctypes.byref(d)) == -1 and b"rows_dev" in lib.mvptr_last_error()
    d = hip.LayerDesc(2, 40, 128, 2, 512, 1e-12, 1, 0, 0, 0, 50, 60, 4096, 4096, 4096)
    assert lib.mvptr_layer_saved_bytes(ctypes.byref(d)) == -1 and b"M_plan" in lib.mvptr_last_error()
    d = hip.LayerDesc(2, 40, 128, 2, 512, 1e-12, 1, 0, 0, 0, 50, 30, 4096, 4096, 4096)
    assert 0 < lib.mvptr_layer_saved_bytes(ctypes.byref(d)) < dense


def test_struct_sizes_match_header():
    from mvp_pytorch_amd import hip
    assert ctypes.sizeof(hip.Dropout) == 16
    assert ctypes.sizeof(hip.LayerWeights) == 16 * 8
    assert ctypes.sizeof(hip.LayerGrads) == 12 * 8
    assert ctypes.sizeof(hip.LayerDesc) == 88      # ABI 5: + stash_bf16 (format of the gelu' stash); ABI 4: + rows_dev, pad_ became M_plan
