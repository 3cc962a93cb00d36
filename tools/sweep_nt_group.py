#!/usr/bin/env python
"""Tile order of the default gemm_nt kernel (MVPTR_NT_GROUP=gm,gn) against time, per shape of the
encoder step; under rocprofv3 --pmc FETCH_SIZE the same launches give the bytes fetched beyond L2:
    sweep_nt_group.py [M] [gm,gn;gm,gn;...]"""
import os
os.environ.setdefault("MVPTR_LIB", "diag")   # kernel-configuration knobs live in the diagnostic build only (make -C mvp_pytorch_amd/csrc diag)
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mvp_pytorch_amd import hip  # noqa: E402

dev = torch.device("cuda:0")
M = int(sys.argv[1]) if len(sys.argv) > 1 else 37748
ORDERS = sys.argv[2].split(";") if len(sys.argv) > 2 else ["4,0", "2,0", "8,0", "4,6", "4,4", "8,4", "8,6", "16,3", "16,2"]
PMC = os.environ.get("PMC_ONCE") == "1"     # one launch per variant (counter passes)


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


SHAPES = [("ffn1 gelu", 3072, 768, hip.EPI_BIAS_GELU), ("qkv bias", 2304, 768, hip.EPI_BIAS), ("ffn2 dgrad", 3072, 768, hip.EPI_GELU_BWD),
          ("o-proj resid", 768, 768, hip.EPI_BIAS_RESID), ("ffn2 resid", 768, 3072, hip.EPI_BIAS_RESID), ("qkv dgrad", 768, 2304, hip.EPI_BIAS)]
for name, N, K, epi in SHAPES:
    a = (torch.randn(M, K, device=dev) * 0.5).to(torch.bfloat16)
    b = (torch.randn(N, K, device=dev) * 0.5).to(torch.bfloat16)
    bias = torch.zeros(N, device=dev)
    out = torch.empty(M, N, device=dev, dtype=torch.uint8 if epi == hip.EPI_BIAS_GELU else torch.bfloat16)
    out1 = torch.empty(M, N, device=dev, dtype=torch.bfloat16) if epi == hip.EPI_BIAS_GELU else None
    aux = torch.randn(M, N, device=dev).to(torch.bfloat16) if epi == hip.EPI_BIAS_RESID else None
    if epi == hip.EPI_GELU_BWD:
        aux = torch.randint(0, 256, (M, N), device=dev, dtype=torch.uint8)
    vec = torch.zeros(N, device=dev) if epi == hip.EPI_GELU_BWD else None
    line = "M=%d %-12s N=%4d K=%4d" % (M, name, N, K)
    for o in ORDERS:
        exp = "0"
        if o.endswith("x"):          # "gm,gnx": without the XCD remap (MVPTR_NT_EXP bit 14)
            o, exp = o[:-1], "16384"
        hip.set_knob("MVPTR_NT_EXP", exp)
        hip.set_knob("MVPTR_NT_GROUP", o)
        fn = lambda: hip.gemm_nt(a, b, epi, bias=bias, aux=aux, out=out, out1=out1, vec_out=vec)  # noqa: E731
        if PMC:
            fn()
            torch.cuda.synchronize()
            continue
        us = min(timeit(fn) for _ in range(3))
        line += "  [%s%s] %6.1f us %4.0f TF" % (o, "x" if exp != "0" else "", us, 2.0 * M * N * K / us / 1e6)
    print(line, flush=True)
hip.set_knob("MVPTR_NT_GROUP", "0,0")
hip.set_knob("MVPTR_NT_EXP", "0")
