#!/usr/bin/env python
"""Round 6: the FFN1 forward GEMM with the 8-bit and with the bf16 gelu' stash (and the plain bias epilogue on the same shape) for
rocprofv3 --pmc passes: how many vector instructions does each epilogue really issue, how long do the waves wait?
    rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d OUT -- python3 tools/pmc_epi.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mvp_pytorch_amd import hip  # noqa: E402

dev = torch.device("cuda:0")
M, H, I = 37748, 768, 3072
x = (torch.randn(M, H, device=dev) * 0.5).to(torch.bfloat16)
w = (torch.randn(I, H, device=dev) * 0.05).to(torch.bfloat16)
b = torch.randn(I, device=dev)
for _ in range(3):
    hip.gemm_nt(x, w, hip.EPI_BIAS, bias=b)
    hip.gemm_nt(x, w, hip.EPI_BIAS_GELU, bias=b)
    hip.gemm_nt(x, w, hip.EPI_BIAS_GELU_BF16, bias=b)
    stash = torch.randint(0, 256, (M, I), device=dev, dtype=torch.uint8)
    hip.gemm_nt(x, w, hip.EPI_GELU_BWD, aux=stash)
torch.cuda.synchronize()
