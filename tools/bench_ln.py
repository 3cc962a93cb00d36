#!/usr/bin/env python
"""LayerNorm forward/backward throughput at the step's row counts (HBM-bound kernels)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mvp_pytorch_amd import hip  # noqa: E402

dev = torch.device("cuda:0")
H = 768


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


for M in (37748, 10917):
    z = torch.randn(M, H, device=dev).to(torch.bfloat16)
    dy = torch.randn(M, H, device=dev).to(torch.bfloat16)
    g, b = torch.rand(H, device=dev) + 0.5, torch.randn(H, device=dev)
    y, mean, rstd = hip.layernorm_fwd(z, g, b, 1e-12)
    dg, db, dbias = (torch.zeros(H, device=dev) for _ in range(3))
    drop = hip.make_dropout(0.1, 1234)
    t_f = timeit(lambda: hip.layernorm_fwd(z, g, b, 1e-12))
    t_fd = timeit(lambda: hip.layernorm_fwd(z, g, b, 1e-12, drop=drop))
    t_b = timeit(lambda: hip.layernorm_bwd(dy, z, mean, rstd, g, dg, db, dbias))
    t_bd = timeit(lambda: hip.layernorm_bwd(dy, z, mean, rstd, g, dg, db, dbias, dense_drop=drop))
    drop2 = hip.make_dropout(0.1, 99)
    t_b2 = timeit(lambda: hip.layernorm_bwd(dy, z, mean, rstd, g, dg, db, dbias, y_drop=drop2, dense_drop=drop))
    by = M * H * 2
    print("M=%5d  bwd with both dropouts (output dropout of the layer above re-applied to dy + dense dropout output) %6.1f us (%.2f TB/s)"
          % (M, t_b2, 4 * by / t_b2 / 1e6), flush=True)
    print("M=%5d  fwd %6.1f us (%.2f TB/s) | fwd+dropout %6.1f us (%.2f TB/s) | bwd %6.1f us (%.2f TB/s) | bwd+dense-dropout output %6.1f us (%.2f TB/s)"
          % (M, t_f, 2 * by / t_f / 1e6, t_fd, 2 * by / t_fd / 1e6, t_b, 3 * by / t_b / 1e6, t_bd, 4 * by / t_bd / 1e6), flush=True)
