#!/usr/bin/env python
"""NS-1 (north_star "fused LayerNorm + projection"): what the tile shape such a fusion needs costs.
A LayerNorm in a GEMM epilogue needs the WHOLE output row (768 columns) in one workgroup: the row-owning 128 x 768 tile
(gemm_nt_rowtile_kernel, diagnostic build, MVPTR_GEMM_CFG=n768) against the product's tiles on the two N = 768 GEMMs that
precede a LayerNorm (attention-output projection K = 768, FFN2 K = 3072), each launch behind a 768-MB write (operands from
HBM as inside the step), with full epilogues and loop-only (MVPTR_NT_EXP bit 10), beside the LayerNorm forward kernel the
fusion would remove.  Fusion pays only if  t(row tile) + LN-epilogue work < t(product tile) + t(LayerNorm kernel)."""
import os
os.environ.setdefault("MVPTR_LIB", "diag")
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mvp_pytorch_amd import hip  # noqa: E402

dev = torch.device("cuda:0")
flush = torch.empty(768 << 20, dtype=torch.uint8, device=dev)


def cold(fn, n=8):
    tot = 0.0
    for i in range(n + 2):
        flush.fill_(i)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        if i >= 2:
            tot += e0.elapsed_time(e1)
    return tot / n * 1e3


def rnd(*s):
    return (torch.randn(*s, device=dev) * 0.5).to(torch.bfloat16)


H, I = 768, 3072
for M in (37748, 10917, 64000):
    for name, K in (("attention-output projection", H), ("FFN2", I)):
        a, w, res = rnd(M, K), rnd(H, K), rnd(M, H)
        bias = torch.randn(H, device=dev)
        out = torch.empty(M, H, device=dev, dtype=torch.bfloat16)
        line = "M=%5d K=%4d %-28s" % (M, K, name)
        ref = None
        for cfg in ("", "t256k", "n768"):
            hip.set_knob("MVPTR_GEMM_CFG", cfg)
            for exp, tag in ((0, "full"), (1024, "loop")):
                hip.set_knob("MVPTR_NT_EXP", exp)
                us = cold(lambda: hip.gemm_nt(a, w, hip.EPI_BIAS_RESID, bias=bias, aux=res, out=out))
                line += "  %s/%s %6.1f" % (cfg or "rule", tag, us)
                if exp == 0:
                    if ref is None:
                        ref = out.float().clone()
                    else:
                        err = ((out.float() - ref).norm() / ref.norm()).item()
                        assert err < 4e-3, (cfg, err)
        hip.set_knob("MVPTR_GEMM_CFG", "")
        hip.set_knob("MVPTR_NT_EXP", 0)
        z = rnd(M, H)
        g, b = torch.ones(H, device=dev), torch.zeros(H, device=dev)
        ln = cold(lambda: hip.layernorm_fwd(z, g, b, 1e-12))
        print(line + "  | LayerNorm fwd %5.1f us" % ln, flush=True)
