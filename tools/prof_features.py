#!/usr/bin/env python
"""Replay of the region-feature path of one configs[1] step for rocprofv3: f32 [256*50, 2054] features -> K-padded bf16
(mvptr_cast_pack, row form) -> 2054 -> 768 image-embedding GEMM (oscar/modeling/modeling_vlbert.py:498).  Every launch runs
behind a 768-MB write so the features come from HBM as in the step.  north_star: "coalesced HBM loads of the 2048-d region
features evidenced by rocprof HBM GB/s"."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mvp_pytorch_amd import hip  # noqa: E402

dev = torch.device("cuda:0")
rows, D, H = 256 * 50, 2054, 768
feats = torch.randn(rows, D, device=dev)
fb = torch.empty(rows, 2056, device=dev, dtype=torch.bfloat16)
w = (torch.randn(H, 2056, device=dev) * 0.02).to(torch.bfloat16)
bias = torch.zeros(H, device=dev)
flush = torch.empty(768 << 20, dtype=torch.uint8, device=dev)
for r in range(int(sys.argv[1]) if len(sys.argv) > 1 else 6):
    flush.fill_(r)
    hip.cast_pack(feats, dst=fb)
    hip.gemm_nt(fb, w, hip.EPI_BIAS, bias=bias)
torch.cuda.synchronize()
print("done")
