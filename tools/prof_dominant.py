#!/usr/bin/env python
"""Replay of the dominant kernel's per-step launch mix for rocprofv3 passes: the grouped
weight-gradient launches (gemm_tn_kernel) of the encoder layers — FFN pair (dW_out, dW_i) and
attention pair (dW_o, dW_qkv + b_qkv) at the three row counts of a configs[1] step — plus the
gemm_nt_kernel family (the eight forward / data-gradient GEMMs of a layer at the same row counts).  bench.py times the same mix."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

from mvp_pytorch_amd.synthetic import synthetic_batch  # noqa: E402

dev = torch.device("cuda:0")
dims = dict(B=256, T=70, P=5, G=20, R=50)
# argv[2] == "full": every slot valid; default: the row counts of bench.py's timed batch (seed 1234)
full = len(sys.argv) > 2 and sys.argv[2] == "full"
batch = None if full else synthetic_batch(dims, bench.BASE_CFG, 1234, device=dev)
mix = bench.DominantMix(dev, dims, bench.BASE_CFG, batch)
print("rows per launch group:", mix.Ms)
cold = os.environ.get("PROF_COLD") == "1"      # every launch behind a 768-MB write, as bench.py times them
flush = torch.empty(768 << 20, dtype=torch.uint8, device=dev) if cold else None
for r in range(int(sys.argv[1]) if len(sys.argv) > 1 else 2):
    for fn in [f for f, _ in mix.tn_launches()] + [f for f, _, _ in mix.nt_all_launches()]:
        if cold:
            flush.fill_(r)
        fn()
torch.cuda.synchronize()
print("done")
