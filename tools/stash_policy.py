#!/usr/bin/env python
"""Cache policy of the 8-bit gelu' stash stores of the FFN1 forward epilogue (half-line stores: 64 bytes per row and wave):
non-temporal (product) / plain / sc1 / sc0 sc1 / nt through a buffer store — diagnostic library, cold launches."""
import os
import sys

import torch

os.environ.setdefault("MVPTR_LIB", "diag")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mvp_pytorch_amd import hip  # noqa: E402
from blas_table import cold_us  # noqa: E402

dev = torch.device("cuda:0")
flush = torch.empty(768 << 20, dtype=torch.uint8, device=dev)
for M in (37748, 10917, 64000):
    a = (torch.randn(M, 768, device=dev) * 0.5).to(torch.bfloat16)
    b = (torch.randn(3072, 768, device=dev) * 0.5).to(torch.bfloat16)
    bias = torch.zeros(3072, device=dev)
    out0 = torch.empty(M, 3072, device=dev, dtype=torch.uint8)
    out1 = torch.empty(M, 3072, device=dev, dtype=torch.bfloat16)
    line = "M=%5d ffn1 fwd GELU:" % M
    for name, bits in (("nt (product)", 0), ("plain", 512), ("sc1", 1 << 19), ("sc0sc1", 2 << 19), ("nt buffer", 3 << 19)):
        hip.set_knob("MVPTR_NT_EXP", bits)
        line += "  %s %.1f" % (name, cold_us(lambda: hip.gemm_nt(a, b, hip.EPI_BIAS_GELU, bias=bias, out=out0, out1=out1), flush, 6))
    hip.set_knob("MVPTR_NT_EXP", 0)
    print(line, flush=True)
