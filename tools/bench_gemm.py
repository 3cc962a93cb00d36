#!/usr/bin/env python
"""Micro-benchmark of the GEMM / attention / LayerNorm kernels at the shapes of one
BASELINE configs[1] step (B=256; L = 75 / 70 / 125).  Run on the GPU box."""
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
    return e0.elapsed_time(e1) / reps * 1e3  # us


def rnd(*s):
    return torch.randn(*s, device=dev).to(torch.bfloat16)


def main():
    H, I = 768, 3072
    for M in (64000, 19200):
        x, xi = rnd(M, H), rnd(M, I)
        x3 = rnd(M, 3 * H)
        shapes = [("qkv  fwd  EPI_BIAS", x, rnd(3 * H, H), hip.EPI_BIAS, None),
                  ("out  fwd  EPI_RESID", x, rnd(H, H), hip.EPI_BIAS_RESID, x),
                  ("ffn1 fwd  EPI_GELU", x, rnd(I, H), hip.EPI_BIAS_GELU, None),
                  ("ffn2 fwd  EPI_RESID", xi, rnd(H, I), hip.EPI_BIAS_RESID, x),
                  ("ffn2 dgrad GELU_BWD", x, rnd(I, H), hip.EPI_GELU_BWD, torch.randint(0, 256, (M, I), device=dev, dtype=torch.uint8)),
                  ("ffn1 dgrad EPI_ADD", xi, rnd(H, I), hip.EPI_ADD, x),
                  ("qkv  dgrad EPI_ADD", x3, rnd(H, 3 * H), hip.EPI_ADD, x)]
        for name, a, b, epi, aux in shapes:
            N, K = b.shape
            bias = torch.zeros(N, device=dev)
            out = torch.empty(M, N, device=dev, dtype=torch.uint8 if epi == hip.EPI_BIAS_GELU else torch.bfloat16)
            out1 = torch.empty(M, N, device=dev, dtype=torch.bfloat16) if epi == hip.EPI_BIAS_GELU else None
            vec = torch.zeros(N, device=dev) if epi == hip.EPI_GELU_BWD else None
            us = timeit(lambda: hip.gemm_nt(a, b, epi, bias=bias, aux=aux, out=out, out1=out1, vec_out=vec))
            print("gemm_nt M=%5d N=%4d K=%4d %-22s %8.1f us  %7.1f TFLOP/s" % (M, N, K, name, us, 2.0 * M * N * K / us / 1e6))
        for name, dy, xx in [("w_qkv", x3, x), ("w_o", x, x), ("w_i", xi, x), ("w_out", x, xi)]:
            N, K = dy.shape[1], xx.shape[1]
            dw = torch.zeros(N, K, device=dev)
            cs = torch.zeros(N, device=dev) if name == "w_qkv" else None
            us = timeit(lambda: hip.gemm_tn(dy, xx, dw, colsum=cs))
            print("gemm_tn M=%5d N=%4d K=%4d %-22s %8.1f us  %7.1f TFLOP/s" % (M, N, K, name, us, 2.0 * M * N * K / us / 1e6))
    for L in (75, 70, 125):
        B, heads = 256, 12
        qkv = rnd(B * L, 3 * H)
        mask = torch.zeros(B, L, device=dev)
        ctx, lse = hip.attention_fwd(qkv, mask, B, L, heads)
        dctx = rnd(B * L, H)
        us = timeit(lambda: hip.attention_fwd(qkv, mask, B, L, heads))
        us2 = timeit(lambda: hip.attention_bwd(qkv, mask, ctx, dctx, lse, B, L, heads))
        byt = B * L * 4 * H * 2
        print("attention L=%3d fwd %7.1f us (%.2f TB/s)   bwd %7.1f us" % (L, us, byt / us / 1e6, us2))
    M = 32000
    z = rnd(M, H)
    g, b = torch.ones(H, device=dev), torch.zeros(H, device=dev)
    y, mean, rstd = hip.layernorm_fwd(z, g, b, 1e-12)
    us = timeit(lambda: hip.layernorm_fwd(z, g, b, 1e-12))
    dg, db, dbias = torch.zeros(H, device=dev), torch.zeros(H, device=dev), torch.zeros(H, device=dev)
    us2 = timeit(lambda: hip.layernorm_bwd(z, z, mean, rstd, g, dg, db, dbias))
    print("layernorm M=%d fwd %6.1f us (%.2f TB/s)  bwd %6.1f us (%.2f TB/s)" % (M, us, M * H * 4 / us / 1e6, us2, M * H * 6 / us2 / 1e6))


if __name__ == "__main__":
    main()
