#!/usr/bin/env python
"""Time the persistent NT kernel (MVPTR_GEMM_CFG=qp) in an ablation build of its tile-end conversion:
    python tools/ablate_qp_conv.py [prod|qpexp1|qpexp2|qpexp3]   (make -C mvp_pytorch_amd/csrc libmvptr_hip_qpexpN.so)"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mvp_pytorch_amd import hip  # noqa: E402

which = sys.argv[1] if len(sys.argv) > 1 else "prod"
if which != "prod":
    hip.LIB_PATH = os.path.join(ROOT, "mvp_pytorch_amd", "csrc", "libmvptr_hip_%s.so" % which)
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


line = which
for (M, N, K) in ((64000, 2304, 768), (64000, 768, 768), (64000, 768, 3072)):
    a = (torch.randn(M, K, device=dev) * 0.5).to(torch.bfloat16)
    b = (torch.randn(N, K, device=dev) * 0.5).to(torch.bfloat16)
    bias = torch.zeros(N, device=dev)
    out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    hip.set_knob("MVPTR_GEMM_CFG", "qp")
    us = timeit(lambda: hip.gemm_nt(a, b, hip.EPI_BIAS, bias=bias, out=out))
    line += "  N=%d K=%d %.1fus %.0fTF" % (N, K, us, 2.0 * M * N * K / us / 1e6)
print(line, flush=True)
