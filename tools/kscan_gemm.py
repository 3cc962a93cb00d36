#!/usr/bin/env python
"""gemm_nt time as a function of K at fixed M, N: the intercept of the linear fit is the per-launch
cost that does not scale with the MFMA loop (pipeline fill + epilogue), the slope the loop rate."""
import os
os.environ.setdefault("MVPTR_LIB", "diag")   # kernel-configuration knobs live in the diagnostic build only (make -C mvp_pytorch_amd/csrc diag)
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mvp_pytorch_amd import hip  # noqa: E402

dev = torch.device("cuda:0")
CFGS = sys.argv[1].split(",") if len(sys.argv) > 1 else ["w4", "t256k", "t256g"]


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


SHAPES = [(32000, 2304, hip.EPI_BIAS, "BIAS"), (32000, 3072, hip.EPI_BIAS_GELU, "GELU"),
          (32000, 3072, hip.EPI_GELU_BWD, "GELU_BWD"), (32000, 768, hip.EPI_BIAS_RESID, "RESID")]
if len(sys.argv) > 2:  # small grids: "M1,M2,..." -> BIAS and GELU at those M
    SHAPES = [(int(m), n, e, nm) for m in sys.argv[2].split(",") for n, e, nm in ((2304, hip.EPI_BIAS, "BIAS"), (3072, hip.EPI_BIAS_GELU, "GELU"))]
for M, N, epi, name in SHAPES:
    Ks = [256, 512, 768, 1536, 3072]
    for cfg in CFGS:
        hip.set_knob("MVPTR_GEMM_CFG", cfg)
        ts = []
        for K in Ks:
            a = (torch.randn(M, K, device=dev) * 0.5).to(torch.bfloat16)
            b = (torch.randn(N, K, device=dev) * 0.5).to(torch.bfloat16)
            bias = torch.zeros(N, device=dev)
            aux = (torch.randn(M, N, device=dev)).to(torch.bfloat16) if epi == hip.EPI_BIAS_RESID else None
            if epi == hip.EPI_GELU_BWD:
                aux = torch.randint(0, 256, (M, N), device=dev, dtype=torch.uint8)
            out = torch.empty(M, N, device=dev, dtype=torch.uint8 if epi == hip.EPI_BIAS_GELU else torch.bfloat16)
            out1 = torch.empty(M, N, device=dev, dtype=torch.bfloat16) if epi == hip.EPI_BIAS_GELU else None
            vec = torch.zeros(N, device=dev) if epi == hip.EPI_GELU_BWD else None
            ts.append(timeit(lambda: hip.gemm_nt(a, b, epi, bias=bias, aux=aux, out=out, out1=out1, vec_out=vec)))
        slope, icpt = np.polyfit(Ks, ts, 1)
        print("%-8s M=%d N=%d cfg=%-6s us@K=%s: %s | fit: %.1f us + %.4f us/K  -> loop rate %.0f TF, fixed part = %.0f%% of the K=768 time"
              % (name, M, N, cfg, Ks, " ".join("%.1f" % t for t in ts), icpt, slope, 2.0 * M * N / slope / 1e6, 100 * icpt / ts[2]), flush=True)
hip.set_knob("MVPTR_GEMM_CFG", "")
