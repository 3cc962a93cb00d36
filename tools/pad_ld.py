#!/usr/bin/env python
"""Does the leading dimension of the OUTPUT (and of the aux operand) matter?  256-row tiles of an [M, N] bf16 matrix
start 256 * N * 2 bytes apart — 1.5 MiB for N = 3072, a multiple of 2^19 — so the tiles written at the same time by
different CUs share their low address bits (HBM channel camping).  Times each GEMM of the step with ld = N and
ld = N + pad."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mvp_pytorch_amd import hip  # noqa: E402

dev = torch.device("cuda:0")
M = int(sys.argv[1]) if len(sys.argv) > 1 else 37748
PADS = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [0, 64, 128, 192, 320]


def timeit(fn, reps=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def mat(rows, cols, pad, dtype=torch.bfloat16, rand=False):
    t = (torch.randn(rows, cols + pad, device=dev) * 0.5).to(dtype) if rand else torch.empty(rows, cols + pad, device=dev, dtype=dtype)
    return t[:, :cols]


SHAPES = [("ffn1 gelu", 3072, 768, hip.EPI_BIAS_GELU), ("qkv bias", 2304, 768, hip.EPI_BIAS), ("ffn2 dgrad", 3072, 768, hip.EPI_GELU_BWD),
          ("o-proj resid", 768, 768, hip.EPI_BIAS_RESID), ("ffn2 resid", 768, 3072, hip.EPI_BIAS_RESID), ("qkv dgrad", 768, 2304, hip.EPI_ADD)]
for name, N, K, epi in SHAPES:
    b = (torch.randn(N, K, device=dev) * 0.5).to(torch.bfloat16)
    bias = torch.zeros(N, device=dev)
    line = "M=%d %-12s N=%4d K=%4d" % (M, name, N, K)
    for pad in PADS:
        a = mat(M, K, pad, rand=True)                # the A operand is an activation too: same padding rule
        out = mat(M, N, pad)
        out1 = mat(M, N, pad) if epi == hip.EPI_BIAS_GELU else None
        aux = mat(M, N, pad, rand=True) if epi in (hip.EPI_BIAS_RESID, hip.EPI_GELU_BWD, hip.EPI_ADD) else None
        vec = torch.zeros(N, device=dev) if epi == hip.EPI_GELU_BWD else None
        fn = lambda: hip.gemm_nt(a, b, epi, bias=None if epi in (hip.EPI_GELU_BWD, hip.EPI_ADD) else bias, aux=aux, out=out, out1=out1, vec_out=vec)  # noqa: E731
        us = min(timeit(fn) for _ in range(3))
        line += "  [ld+%d] %6.1f us %4.0f TF" % (pad, us, 2.0 * M * N * K / us / 1e6)
    print(line, flush=True)
