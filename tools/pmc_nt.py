#!/usr/bin/env python
"""One launch per gemm_nt configuration and shape for rocprofv3 --pmc passes (L2 hit rate, traffic):
    rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d DIR -- python3 tools/pmc_nt.py [cfgs]"""
import os
os.environ.setdefault("MVPTR_LIB", "diag")   # kernel-configuration knobs live in the diagnostic build only (make -C mvp_pytorch_amd/csrc diag)
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mvp_pytorch_amd import hip  # noqa: E402

dev = torch.device("cuda:0")
cfgs = sys.argv[1].split(",") if len(sys.argv) > 1 else ["t256k", "q"]
M = int(sys.argv[2]) if len(sys.argv) > 2 else 64000
for (N, K, epi) in ((2304, 768, hip.EPI_BIAS), (3072, 768, hip.EPI_BIAS), (768, 768, hip.EPI_BIAS), (768, 3072, hip.EPI_BIAS)):
    a = (torch.randn(M, K, device=dev) * 0.5).to(torch.bfloat16)
    b = (torch.randn(N, K, device=dev) * 0.5).to(torch.bfloat16)
    bias = torch.zeros(N, device=dev)
    out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    for cfg in cfgs:
        hip.set_knob("MVPTR_GEMM_CFG", cfg)
        for _ in range(2):
            hip.gemm_nt(a, b, epi, bias=bias, out=out)
torch.cuda.synchronize()
print("done")
