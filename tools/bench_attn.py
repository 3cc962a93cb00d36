#!/usr/bin/env python
"""Attention forward / backward timings with and without dropout at the step's shapes."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mvp_pytorch_amd import hip  # noqa: E402

dev = torch.device("cuda:0")


def timeit(fn, reps=20):
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


H, heads = 768, 12
for B, L in ((512, 125), (256, 75), (256, 70)):
    qkv = torch.randn(B * L, 3 * H, device=dev).to(torch.bfloat16)
    mask = torch.zeros(B, L, device=dev)
    dctx = torch.randn(B * L, H, device=dev).to(torch.bfloat16)
    drop = hip.make_dropout(0.1, 99)
    out = []
    for d in (None, drop):
        ctx, lse = hip.attention_fwd(qkv, mask, B, L, heads, drop=d)
        tf = timeit(lambda: hip.attention_fwd(qkv, mask, B, L, heads, drop=d))
        tb = timeit(lambda: hip.attention_bwd(qkv, mask, ctx, dctx, lse, B, L, heads, drop=d))
        out.append((tf, tb))
    byf, byb = B * L * 4 * H * 2, B * L * (3 + 1 + 1 + 3) * H * 2
    print("B=%d L=%d: fwd %.1f us (%.2f TB/s), with dropout %.1f us | bwd %.1f us (%.2f TB/s), with dropout %.1f us"
          % (B, L, out[0][0], byf / out[0][0] / 1e6, out[1][0], out[0][1], byb / out[0][1] / 1e6, out[1][1]))
