#!/usr/bin/env python
"""Time every gemm_nt tile configuration (MVPTR_GEMM_CFG) at the shapes of a configs[1] step and
check each against the default configuration's output.  Run on the GPU box."""
import os
os.environ.setdefault("MVPTR_LIB", "diag")   # kernel-configuration knobs live in the diagnostic build only (make -C mvp_pytorch_amd/csrc diag)
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mvp_pytorch_amd import hip  # noqa: E402

dev = torch.device("cuda:0")
CFGS = sys.argv[1].split(",") if len(sys.argv) > 1 else ["w4", "w4g", "t256", "t256g", "t256k"]


COLD = os.environ.get("SWEEP_COLD") == "1"   # every timed launch behind a 768-MB write: operands come from HBM as inside the training step
_flush = torch.empty(768 << 20, dtype=torch.uint8, device=dev) if COLD else None


def timeit(fn, reps=20):
    if COLD:
        tot = 0.0
        for i in range(8):
            _flush.fill_(i)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            fn()
            e1.record()
            torch.cuda.synchronize()
            if i >= 2:
                tot += e0.elapsed_time(e1)
        return tot / 6 * 1e3
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


def rnd(*s):
    return (torch.randn(*s, device=dev) * 0.5).to(torch.bfloat16)


H, I = 768, 3072
MS = [int(v) for v in sys.argv[2].split(",")] if len(sys.argv) > 2 else [37748, 10917, 3000]
for M in MS:
    x, xi, x3 = rnd(M, H), rnd(M, I), rnd(M, 3 * H)
    shapes = [("qkv fwd BIAS", x, rnd(3 * H, H), hip.EPI_BIAS, None),
              ("out fwd RESID", x, rnd(H, H), hip.EPI_BIAS_RESID, x),
              ("ffn1 fwd GELU", x, rnd(I, H), hip.EPI_BIAS_GELU, None),
              ("ffn2 fwd RESID", xi, rnd(H, I), hip.EPI_BIAS_RESID, x),
              ("ffn2 dgrad GELU_BWD", x, rnd(I, H), hip.EPI_GELU_BWD, xi),
              ("ffn1 dgrad ADD", xi, rnd(H, I), hip.EPI_ADD, x),
              ("qkv dgrad ADD", x3, rnd(H, 3 * H), hip.EPI_ADD, x)]
    for name, a, b, epi, aux in shapes:
        N, K = b.shape
        bias = torch.zeros(N, device=dev)
        out1 = torch.empty(M, N, device=dev, dtype=torch.bfloat16) if epi == hip.EPI_BIAS_GELU else None
        hip.set_knob("MVPTR_GEMM_CFG", "")
        ref = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
        hip.gemm_nt(a, b, epi, bias=bias, aux=aux, out=ref, out1=out1,
                    vec_out=torch.zeros(N, device=dev) if epi == hip.EPI_GELU_BWD else None)
        ref = ref.float()
        line = "M=%5d N=%4d K=%4d %-20s" % (M, N, K, name)
        for cfg in CFGS:
            hip.set_knob("MVPTR_GEMM_CFG", cfg)
            out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
            vec = torch.zeros(N, device=dev) if (epi == hip.EPI_GELU_BWD and cfg not in ("q", "qp")) else None
            us = timeit(lambda: hip.gemm_nt(a, b, epi, bias=bias, aux=aux, out=out, out1=out1, vec_out=vec))
            ok = torch.equal(out.float(), ref) or ((out.float() - ref).norm() / ref.norm()).item() < 4e-3
            line += "  %s %6.1fus %6.1fTF%s" % (cfg, us, 2.0 * M * N * K / us / 1e6, "" if ok else " MISMATCH")
        print(line, flush=True)
hip.set_knob("MVPTR_GEMM_CFG", "")
