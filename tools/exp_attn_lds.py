#!/usr/bin/env python
"""Experiment: packed attention where every sequence is short (60 rows) but the LDS tiles are sized for
the longest sequence of the batch (125) vs sized for 64: what does LDS-limited occupancy cost?"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mvp_pytorch_amd import hip  # noqa: E402

dev = torch.device("cuda:0")
B, heads, H = 512, 12, 768


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


for L in (30, 60, 90):
    lens = torch.full((B,), L, dtype=torch.int32, device=dev)
    starts = (torch.arange(B, device=dev, dtype=torch.int32) * L).contiguous()
    qkv = torch.randn(B * L, 3 * H, device=dev).to(torch.bfloat16)
    dctx = torch.randn(B * L, H, device=dev).to(torch.bfloat16)
    line = "all sequences %d rows:" % L
    for Lmax in (125, (L + 31) // 32 * 32):
        ctx, lse = hip.attention_fwd_packed(qkv, starts, lens, B, Lmax, heads)
        tf = timeit(lambda: hip.attention_fwd_packed(qkv, starts, lens, B, Lmax, heads))
        tb = timeit(lambda: hip.attention_bwd_packed(qkv, starts, lens, ctx, dctx, lse, B, Lmax, heads))
        line += "  tiles for %3d rows: fwd %.1f us bwd %.1f us" % (Lmax, tf, tb)
    print(line)
