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
for B, L in ((512, 125), (512, 96), (512, 64), (256, 75), (256, 70)):
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

# the packed launches of the timed batch (bench.py's synthetic batch): text, visual and joint stacks
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from mvp_pytorch_amd.synthetic import synthetic_batch  # noqa: E402

dims = dict(B=256, T=70, P=5, G=20, R=50)
b = synthetic_batch(dims, bench.BASE_CFG, 1234, device=dev)
la, lb = b["input_mask_a"].sum(1).int(), b["input_mask_b"].sum(1).int()
lb_cut = (b["input_mask_b"][:, dims["G"]:].sum(1)).int()
stacks = {"text": la, "visual": lb, "joint (matched + hard pairs, lengths of the matched ones twice)": torch.cat([la + lb_cut, la + lb_cut])}
for name, lens in stacks.items():
    n = lens.numel()
    start = (torch.cumsum(lens, 0) - lens).int()
    rows, lmax = int(lens.sum()), int(lens.max())
    qkv = torch.randn(rows, 3 * H, device=dev).to(torch.bfloat16)
    dctx = torch.randn(rows, H, device=dev).to(torch.bfloat16)
    drop = hip.make_dropout(0.1, 99)
    ctx, lse = hip.attention_fwd_packed(qkv, start, lens, n, lmax, heads, drop=drop)
    tf = timeit(lambda: hip.attention_fwd_packed(qkv, start, lens, n, lmax, heads, drop=drop))
    tb = timeit(lambda: hip.attention_bwd_packed(qkv, start, lens, ctx, dctx, lse, n, lmax, heads, drop=drop))
    fl = float((4.0 * lens.double() ** 2 * H).sum())           # QK^T + PV forward; backward 2.5x
    print("packed %s: %d sequences, %d rows, longest %d: fwd %.1f us (%.2f TB/s, %.0f TFLOP/s) | bwd %.1f us (%.2f TB/s, %.0f TFLOP/s), both with dropout"
          % (name, n, rows, lmax, tf, rows * 4 * H * 2 / tf / 1e6, fl / tf / 1e6, tb, rows * 8 * H * 2 / tb / 1e6, 2.5 * fl / tb / 1e6))
