#!/usr/bin/env python
"""Do the ROW STRIDES of the GEMM operands matter (L2 / HBM channel camping)?  A k-step of the 256 x 256 tile reads 128 bytes
from each of 512 rows; with K = 768 the rows are 1536 bytes apart (12 lines: 4 of 16 channels under a modulo interleave),
with K = 3072 6144 bytes (every row on the same channel).  Times each GEMM with the A (activation), B (weight) and
output / aux leading dimensions padded separately and together, cold operands; diagnostic library: also loop-only."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mvp_pytorch_amd import hip  # noqa: E402
from blas_table import cold_us  # noqa: E402

dev = torch.device("cuda:0")
M = int(sys.argv[1]) if len(sys.argv) > 1 else 37748
PAD = int(sys.argv[2]) if len(sys.argv) > 2 else 64
diag = os.environ.get("MVPTR_LIB") == "diag"
flush = torch.empty(768 << 20, dtype=torch.uint8, device=dev)


def mat(rows, cols, pad, rand=True, dtype=torch.bfloat16):
    if dtype == torch.uint8:
        t = torch.randint(0, 256, (rows, cols + pad), device=dev, dtype=torch.uint8)
    else:
        t = (torch.randn(rows, cols + pad, device=dev) * 0.5).to(dtype) if rand else torch.empty(rows, cols + pad, device=dev, dtype=dtype)
    return t[:, :cols]


SHAPES = [("qkv fwd BIAS", 2304, 768, hip.EPI_BIAS), ("out fwd RESID", 768, 768, hip.EPI_BIAS_RESID), ("ffn1 fwd GELU", 3072, 768, hip.EPI_BIAS_GELU),
          ("ffn2 fwd RESID", 768, 3072, hip.EPI_BIAS_RESID), ("ffn2 dgrad GELU_BWD", 3072, 768, hip.EPI_GELU_BWD), ("qkv dgrad ADD", 768, 2304, hip.EPI_ADD)]
for name, N, K, epi in SHAPES:
    bias = torch.zeros(N, device=dev)
    line = "M=%d %-20s N=%4d K=%4d" % (M, name, N, K)
    for label, pa, pb, pc in (("none", 0, 0, 0), ("A", PAD, 0, 0), ("B", 0, PAD, 0), ("A+B", PAD, PAD, 0), ("C", 0, 0, PAD), ("all", PAD, PAD, PAD)):
        a, b = mat(M, K, pa), mat(N, K, pb)
        out = mat(M, N, pc, rand=False, dtype=torch.uint8 if epi == hip.EPI_BIAS_GELU else torch.bfloat16)
        out1 = mat(M, N, pc, rand=False) if epi == hip.EPI_BIAS_GELU else None
        aux = None
        if epi in (hip.EPI_BIAS_RESID, hip.EPI_ADD):
            aux = mat(M, N, pc)
        elif epi == hip.EPI_GELU_BWD:
            aux = mat(M, N, pc, dtype=torch.uint8)
        vec = torch.zeros(N, device=dev) if epi == hip.EPI_GELU_BWD else None
        fn = lambda: hip.gemm_nt(a, b, epi, bias=None if epi in (hip.EPI_GELU_BWD, hip.EPI_ADD) else bias, aux=aux, out=out, out1=out1, vec_out=vec)  # noqa: E731
        us = cold_us(fn, flush, 4)
        line += "  %s %.1f" % (label, us)
        if diag and label in ("none", "A+B"):
            hip.set_knob("MVPTR_NT_EXP", 1024)
            line += " (loop %.1f)" % cold_us(fn, flush, 4)
            hip.set_knob("MVPTR_NT_EXP", 0)
    print(line, flush=True)
