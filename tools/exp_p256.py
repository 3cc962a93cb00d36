#!/usr/bin/env python
"""A/B of the persistent GEMM with and without next-tile prefetch vs the one-tile-per-workgroup kernel."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mvp_pytorch_amd import hip  # noqa: E402

dev = torch.device("cuda:0")


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


M = 64000
for N, K, epi, name in ((2304, 768, hip.EPI_BIAS, "BIAS"), (3072, 768, hip.EPI_BIAS_GELU, "GELU"), (768, 3072, hip.EPI_ADD, "ADD K=3072"), (768, 768, hip.EPI_ADD, "ADD K=768")):
    a = (torch.randn(M, K, device=dev) * 0.5).to(torch.bfloat16)
    b = (torch.randn(N, K, device=dev) * 0.5).to(torch.bfloat16)
    bias = torch.zeros(N, device=dev)
    out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    out1 = torch.empty(M, N, device=dev, dtype=torch.bfloat16) if epi == hip.EPI_BIAS_GELU else None
    res = {}
    for rep in range(3):
        for cfg, xp in (("t256k", "0"), ("p256", "0"), ("p256", "1")):
            hip.set_knob("MVPTR_GEMM_CFG", cfg)
            hip.set_knob("MVPTR_NT_EXP", xp)
            res.setdefault((cfg, xp), []).append(timeit(lambda: hip.gemm_nt(a, b, epi, bias=bias, out=out, out1=out1)))
    print("%-12s M=%d N=%d K=%d: " % (name, M, N, K) + "  ".join("%s/%s min %.1f" % (k[0], "noprefetch" if k[1] == "1" else "std", min(v)) for k, v in res.items()), flush=True)
