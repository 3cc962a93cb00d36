#!/usr/bin/env python
"""Diagnostic: per-segment cycle shares of the gemm_nt main loop from in-kernel s_memtime stamps
(build: hipcc -DMVPTR_STAMP_BUILD ... -> mvp_pytorch_amd/csrc/libmvptr_hip_stamp.so).  The stamped
build fences overlaps the real kernel has: read its SHARES, not its run time."""
import ctypes
import os
os.environ.setdefault("MVPTR_LIB", "diag")   # kernel-configuration knobs live in the diagnostic build only (make -C mvp_pytorch_amd/csrc diag)
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mvp_pytorch_amd import hip  # noqa: E402

hip.LIB_PATH = os.path.join(ROOT, "mvp_pytorch_amd", "csrc", "libmvptr_hip_stamp.so")
dev = torch.device("cuda:0")


def run(M, N, K, epi, name):
    a = torch.randn(M, K, device=dev).to(torch.bfloat16)
    b = torch.randn(N, K, device=dev).to(torch.bfloat16)
    bias = torch.zeros(N, device=dev)
    aux = torch.randn(M, N, device=dev).to(torch.bfloat16) if epi in (hip.EPI_BIAS_RESID,) else None
    nwg = ((M + 255) // 256) * ((N + 127) // 128) * 2
    st = torch.zeros(nwg * 8, dtype=torch.int64, device=dev)
    hip.set_knob("MVPTR_GEMM_STAMPS", str(st.data_ptr()))
    out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    out1 = torch.empty(M, N, device=dev, dtype=torch.bfloat16) if epi == hip.EPI_BIAS_GELU else None
    for _ in range(3):
        hip.gemm_nt(a, b, epi, bias=bias, aux=aux, out=out, out1=out1)
    torch.cuda.synchronize()
    s = st.view(nwg, 8).double().cpu()
    nk = s[:, 4].mean().item()
    w, i, l, m = (s[:, j].mean().item() / nk for j in range(4))
    tot = w + i + l + m
    print("%-28s M=%d N=%d K=%d  cycles per K-step (wave 0): wait(vmcnt+barrier) %.0f  issue(ds_read+lds-dma) %.0f  lds-wait %.0f  mfma %.0f  | total %.0f  (mfma share %.0f%%)"
          % (name, M, N, K, w, i, l, m, tot, 100 * m / tot))


hip.set_knob("MVPTR_GEMM_CFG", "w4")
run(32000, 768, 3072, hip.EPI_BIAS_RESID, "ffn2 fwd (w4)")
run(32000, 2304, 768, hip.EPI_BIAS, "qkv fwd (w4)")
run(32000, 3072, 768, hip.EPI_BIAS_GELU, "ffn1 fwd (w4)")
for cfg in ("t256k", "t256"):
    hip.set_knob("MVPTR_GEMM_CFG", cfg)
    run(19200, 768, 3072, hip.EPI_BIAS_RESID, "ffn2 fwd (%s)" % cfg)
