#!/usr/bin/env python
"""Speed experiment: gemm_tn with the experimental library (results are NOT valid)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mvp_pytorch_amd import hip  # noqa: E402

# argv[1]: "prod" | "exp" (one 16-B read per fragment) | "exp2" no MFMA | "exp3" no LDS reads |
# "exp4" no LDS-DMA | "exp5" no atomic write-out   (make -C mvp_pytorch_amd/csrc ablate)
if len(sys.argv) > 1 and sys.argv[1].startswith("exp"):
    hip.LIB_PATH = os.path.join(ROOT, "mvp_pytorch_amd", "csrc", "libmvptr_hip_%s.so" % sys.argv[1])
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
for N, K, name in ((3072, 768, "w_i"), (768, 3072, "w_out"), (2304, 768, "w_qkv")):
    dy = (torch.randn(M, N, device=dev) * 0.5).to(torch.bfloat16)
    x = (torch.randn(M, K, device=dev) * 0.5).to(torch.bfloat16)
    dw = torch.zeros(N, K, device=dev)
    line = "%s %-6s" % (sys.argv[1] if len(sys.argv) > 1 else "prod", name)
    for cfg in ("32", "K", "k2"):
        hip.set_knob("MVPTR_GEMM_TN", cfg)
        us = min(timeit(lambda: hip.gemm_tn(dy, x, dw)) for _ in range(3))
        line += "  %s %6.1fus %5.0fTF" % (cfg, us, 2.0 * M * N * K / us / 1e6)
    print(line, flush=True)
