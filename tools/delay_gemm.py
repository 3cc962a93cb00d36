#!/usr/bin/env python
"""Experiment: start the second resident workgroup of every CU late (MVPTR_GEMM_DELAY) so co-resident
workgroups stop running their epilogues (HBM bursts) at the same time."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mvp_pytorch_amd import hip  # noqa: E402

dev = torch.device("cuda:0")


def timeit(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


hip.set_knob("MVPTR_GEMM_CFG", "t256k")
for M in (64000, 32000):
    for N, K, epi, name in [(2304, 768, hip.EPI_BIAS, "BIAS"), (3072, 768, hip.EPI_BIAS_GELU, "GELU"),
                            (3072, 768, hip.EPI_GELU_BWD, "GELU_BWD"), (768, 3072, hip.EPI_BIAS_RESID, "RESID")]:
        a = (torch.randn(M, K, device=dev) * 0.5).to(torch.bfloat16)
        b = (torch.randn(N, K, device=dev) * 0.5).to(torch.bfloat16)
        bias = torch.zeros(N, device=dev)
        aux = (torch.randn(M, N, device=dev)).to(torch.bfloat16) if epi in (hip.EPI_GELU_BWD, hip.EPI_BIAS_RESID) else None
        out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
        out1 = torch.empty(M, N, device=dev, dtype=torch.bfloat16) if epi == hip.EPI_BIAS_GELU else None
        vec = torch.zeros(N, device=dev) if epi == hip.EPI_GELU_BWD else None
        line = "%-8s M=%d N=%d K=%d:" % (name, M, N, K)
        settings = tuple(sys.argv[1].split(";")) if len(sys.argv) > 1 else ("0", "60000,256,-8", "40000,256,-4", "30000,256,-8", "0")
        best = {}
        for rep in range(3):  # interleave the settings so clock / cache state is shared
            for d in settings:
                hip.set_knob("MVPTR_GEMM_DELAY", d)
                us = timeit(lambda: hip.gemm_nt(a, b, epi, bias=bias, aux=aux, out=out, out1=out1, vec_out=vec), reps=10)
                best.setdefault(d, []).append(us)
        for d in dict.fromkeys(settings):
            line += "  [%s] min %.1f med %.1f" % (d, min(best[d]), sorted(best[d])[len(best[d]) // 2])
        print(line, flush=True)
