#!/usr/bin/env python
"""Grouped weight-gradient launches (FFN pair, attention pair of an encoder layer) per gemm_tn
configuration and forced M-split count: tools/sweep_tn_group.py [cfgs] [splits]."""
import os
os.environ.setdefault("MVPTR_LIB", "diag")   # kernel-configuration knobs live in the diagnostic build only (make -C mvp_pytorch_amd/csrc diag)
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mvp_pytorch_amd import hip  # noqa: E402

if os.environ.get("MVPTR_TOOL_LIB"):     # ablation build (make -C mvp_pytorch_amd/csrc libmvptr_hip_exp4.so ...): speed only
    hip.LIB_PATH = os.path.join(os.path.dirname(hip.LIB_PATH), os.environ["MVPTR_TOOL_LIB"])
dev = torch.device("cuda:0")
CFGS = sys.argv[1].split(",") if len(sys.argv) > 1 else ["auto", "32", "q"]
SPLITS = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [0]
MS = [int(x) for x in sys.argv[3].split(",")] if len(sys.argv) > 3 else [64000, 37748, 19200, 10917]
ORDERS = [int(x) for x in sys.argv[4].split(",")] if len(sys.argv) > 4 else [0]


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


GROUPS = {"ffn": [(768, 3072), (3072, 768)], "attn": [(2304, 768), (768, 768)]}
for M in MS:
    for gname, shapes in GROUPS.items():
        probs = []
        for N, K in shapes:
            dy = (torch.randn(M, N, device=dev) * 0.5).to(torch.bfloat16)
            x = (torch.randn(M, K, device=dev) * 0.5).to(torch.bfloat16)
            probs.append((dy, x, torch.zeros(N, K, device=dev), None))
        flops = sum(2.0 * M * N * K for N, K in shapes)
        line = "M=%5d %-4s" % (M, gname)
        for cfg in CFGS:
            hip.set_knob("MVPTR_GEMM_TN", "" if cfg == "auto" else cfg)
            for sp in SPLITS:
                hip.set_knob("MVPTR_TN_SPLITS", str(sp))
                for order in ORDERS:
                    hip.set_knob("MVPTR_NT_EXP", str(order))
                    us = min(timeit(lambda: hip.gemm_tn_multi(probs)) for _ in range(3))
                    line += "  %s/%d/o%d %6.1fus %5.0fTF" % (cfg, sp, order, us, flops / us / 1e6)
                hip.set_knob("MVPTR_NT_EXP", "0")
        print(line, flush=True)
hip.set_knob("MVPTR_GEMM_TN", "")
hip.set_knob("MVPTR_TN_SPLITS", "0")
