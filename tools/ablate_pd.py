#!/usr/bin/env python
"""pd kernel (persistent, deferred epilogue, two waves per SIMD) with and without its deferred work
(MVPTR_NT_EXP bit 7) against the default kernel and the no-epilogue Q kernel."""
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


for (M, N, K) in ((64000, 2304, 768), (64128, 2304, 768), (64000, 768, 768), (64000, 768, 3072)):
    a = (torch.randn(M, K, device=dev) * 0.5).to(torch.bfloat16)
    b = (torch.randn(N, K, device=dev) * 0.5).to(torch.bfloat16)
    bias = torch.zeros(N, device=dev)
    out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    line = "M=%d N=%d K=%d BIAS" % (M, N, K)
    for cfg, exp in (("t256k", 0), ("pd", 0), ("pd", 512), ("pd", 128), ("pd", 0), ("pd", 512)):
        hip.set_knob("MVPTR_GEMM_CFG", cfg)
        hip.set_knob("MVPTR_NT_EXP", str(exp))
        us = min(timeit(lambda: hip.gemm_nt(a, b, hip.EPI_BIAS, bias=bias, out=out)) for _ in range(2))
        line += "  %s/%d %.1fus %.0fTF" % (cfg, exp, us, 2.0 * M * N * K / us / 1e6)
    print(line, flush=True)
hip.set_knob("MVPTR_GEMM_CFG", "")
hip.set_knob("MVPTR_NT_EXP", "0")
