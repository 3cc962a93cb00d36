#!/usr/bin/env python
"""Round 6 debug: is the GELU epilogue a function of the element alone?  The same A rows at different row positions / batch sizes."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mvp_pytorch_amd import hip
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
for (N, K) in ((512, 128), (3072, 768)):
    b = (torch.randn(N, K, generator=g) * 0.3).to(torch.bfloat16).to(dev)
    bias = torch.randn(N, generator=g).to(dev)
    rows = (torch.randn(180, K, generator=g) * 1.5).to(torch.bfloat16).to(dev)
    dq0, act0 = hip.gemm_nt(rows, b, hip.EPI_BIAS_GELU, bias=bias)
    for M in (180, 181, 300, 517, 1100):
        for off in (0, 1, 7, 100):
            if off + 180 > M:
                continue
            a = torch.zeros(M, K, dtype=torch.bfloat16, device=dev)
            a[off:off + 180] = rows
            dq, act = hip.gemm_nt(a, b, hip.EPI_BIAS_GELU, bias=bias)
            same_a = torch.equal(act[off:off + 180], act0)
            same_d = torch.equal(dq[off:off + 180], dq0)
            if not (same_a and same_d):
                da = (act[off:off + 180].float() - act0.float()).abs()
                print("N %d K %d M %d off %d: act equal %s (max diff %.3e at %s), stash equal %s (%d bytes differ)" % (
                    N, K, M, off, same_a, float(da.max()), tuple(int(v) for v in torch.nonzero(da == da.max())[0]), same_d,
                    int((dq[off:off + 180] != dq0).sum())))
    print("N %d K %d done" % (N, K))
