#!/usr/bin/env python
"""Diagnostic library only: the persistent gemm_nt kernel with its stores dropped / non-temporal / write-through
(MVPTR_NT_EXP bits 13-15) against the round-3 kernel, cold operands.  What do the OUTPUT WRITES cost these GEMMs?"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MVPTR_LIB", "diag")
from mvp_pytorch_amd import hip  # noqa: E402
from blas_table import cold_us, rnd  # noqa: E402

dev = torch.device("cuda:0")
flush = torch.empty(768 << 20, dtype=torch.uint8, device=dev)
H, I = 768, 3072
for M in (37748, 64000):
    for name, N, K, epi in (("qkv fwd BIAS", 2304, 768, hip.EPI_BIAS), ("ffn2 fwd RESID", 768, 3072, hip.EPI_BIAS_RESID),
                            ("out fwd RESID", 768, 768, hip.EPI_BIAS_RESID), ("ffn1 fwd GELU", 3072, 768, hip.EPI_BIAS_GELU)):
        a, b = rnd(M, K), rnd(N, K)
        bias = torch.zeros(N, device=dev)
        aux = rnd(M, N) if epi == hip.EPI_BIAS_RESID else None
        out = torch.empty(M, N, device=dev, dtype=torch.uint8 if epi == hip.EPI_BIAS_GELU else torch.bfloat16)
        out1 = torch.empty(M, N, device=dev, dtype=torch.bfloat16) if epi == hip.EPI_BIAS_GELU else None
        fn = lambda: hip.gemm_nt(a, b, epi, bias=bias, aux=aux, out=out, out1=out1)  # noqa: E731
        res = []
        for label, cfg, exp in (("t256k", "t256k", 0), ("t256k no-epi", "t256k", 1024), ("P", "p", 0), ("P dropped", "p", 1 << 13), ("P nt", "p", 2 << 13),
                                ("P sc1", "p", 3 << 13), ("P sc0sc1", "p", 4 << 13)):
            hip.set_knob("MVPTR_GEMM_CFG", cfg)
            hip.set_knob("MVPTR_NT_EXP", exp)
            res.append("%s %.1f" % (label, cold_us(fn, flush, 4)))
        hip.set_knob("MVPTR_GEMM_CFG", "")
        hip.set_knob("MVPTR_NT_EXP", 0)
        print("M=%d %-16s N=%d K=%d: %s" % (M, name, N, K, " | ".join(res)), flush=True)
