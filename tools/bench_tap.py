#!/usr/bin/env python
"""mvptr_tap_rows_bwd at the shape of the step's largest call: the gradient of the joint stack's INPUT rows (sized for the sync-free
bound, valid rows first, the rest idx -1) summed into the text + visual output rows that were gathered 1-3 times each."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mvp_pytorch_amd import hip  # noqa: E402

dev = torch.device("cuda:0")
H = 768
g = torch.Generator().manual_seed(0)
for bound, valid, rows, rows2 in ((64000, 37748, 10917, 11143), (37748, 37748, 10917, 11143), (64000, 37748, 22060, 0)):
    R = rows + rows2
    # every destination row once, then random extra uses up to `valid` entries
    idx = torch.cat([torch.randperm(R, generator=g), torch.randint(0, R, (valid - R,), generator=g)])
    idx = idx[torch.randperm(valid, generator=g)]
    idx = torch.cat([idx, torch.full((bound - valid,), -1, dtype=torch.long)]).to(torch.int32).to(dev)
    grad = torch.randn(bound, H, device=dev).to(torch.bfloat16)
    d, d2 = hip.tap_rows_bwd([(grad, idx)], rows, rows2, H)
    ref = torch.zeros(R, H, device=dev).index_add_(0, idx[:valid].long(), grad[:valid].float())
    got = torch.cat([d, d2]) if d2 is not None else d
    err = float((got.float() - ref).abs().max() / ref.abs().max())
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        hip.tap_rows_bwd([(grad, idx)], rows, rows2, H)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 10 * 1e3
    by = (valid + R) * H * 2
    print("bound %d valid %d -> %d + %d rows: %.1f us (%.2f TB/s over entries read + rows written), max rel err %.1e" % (bound, valid, rows, rows2, us, by / us / 1e6, err), flush=True)
