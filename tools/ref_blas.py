#!/usr/bin/env python
"""Reference point only (not used by the product path): what torch.matmul (hipBLASLt) reaches on the
plain GEMM shapes of a step, without any fused epilogue."""
import torch

dev = torch.device("cuda:0")


def bench(fn, reps=20):
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


def rnd(*s):
    return (torch.randn(*s, device=dev) * 0.5).to(torch.bfloat16)


for M in (64000, 19200):
    for N, K in ((2304, 768), (768, 768), (3072, 768), (768, 3072)):
        a, w = rnd(M, K), rnd(N, K)
        us = bench(lambda: torch.matmul(a, w.t()))
        print("nt  M=%5d N=%4d K=%4d  %7.1f us  %6.0f TF" % (M, N, K, us, 2.0 * M * N * K / us / 1e6))
    for N, K in ((2304, 768), (768, 768), (3072, 768), (768, 3072)):
        dy, x = rnd(M, N), rnd(M, K)
        us = bench(lambda: torch.matmul(dy.t(), x))
        print("tn  M=%5d N=%4d K=%4d  %7.1f us  %6.0f TF  (bf16 out)" % (M, N, K, us, 2.0 * M * N * K / us / 1e6))
