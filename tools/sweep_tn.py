#!/usr/bin/env python
"""Time the gemm_tn configurations (MVPTR_GEMM_TN) at the weight-gradient shapes of a step and check
each against an f32 reference."""
import os
os.environ.setdefault("MVPTR_LIB", "diag")   # kernel-configuration knobs live in the diagnostic build only (make -C mvp_pytorch_amd/csrc diag)
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mvp_pytorch_amd import hip  # noqa: E402

dev = torch.device("cuda:0")
CFGS = sys.argv[1].split(",") if len(sys.argv) > 1 else ["auto", "32", "64", "k2", "K", "q"]


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


for M in (64000, 37748, 10917, 3000):
    for N, K, name in ((2304, 768, "w_qkv"), (768, 768, "w_o"), (3072, 768, "w_i"), (768, 3072, "w_out")):
        dy = (torch.randn(M, N, device=dev) * 0.5).to(torch.bfloat16)
        x = (torch.randn(M, K, device=dev) * 0.5).to(torch.bfloat16)
        ref = None
        line = "M=%5d N=%4d K=%4d %-6s" % (M, N, K, name)
        for cfg in CFGS:
            if cfg == "auto":
                hip.set_knob("MVPTR_GEMM_TN", "")
            else:
                hip.set_knob("MVPTR_GEMM_TN", cfg)
            dw = torch.zeros(N, K, device=dev)
            cs = torch.zeros(N, device=dev)
            hip.gemm_tn(dy, x, dw, colsum=cs)
            if ref is None:
                ref = dy[:4096].float().t() @ x[:4096].float() if M > 4096 else dy.float().t() @ x.float()
                refc = dy.float().sum(0)
            if M <= 4096:
                err = ((dw - ref).norm() / ref.norm()).item()
            else:
                err = float("nan")
            errc = ((cs - refc).norm() / refc.norm()).item()
            dw.zero_()
            us = timeit(lambda: hip.gemm_tn(dy, x, dw))
            line += "  %s %6.1fus %5.0fTF" % (cfg, us, 2.0 * M * N * K / us / 1e6)
            if (err == err and err > 1e-3) or errc > 1e-3:
                line += " ERR(%.1e,%.1e)" % (err, errc)
        print(line, flush=True)
hip.set_knob("MVPTR_GEMM_TN", "")
