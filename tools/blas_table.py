#!/usr/bin/env python
"""Cold, in-step-like table of this library's forward / data-gradient GEMMs (with their fused epilogues) against
torch.matmul (hipBLASLt, plain bf16 output, no epilogue) on every GEMM shape of a configs[1] training step
(VERDICT r03 #1d).  Every launch runs behind a 768-MB write, so operands come from HBM as they do inside the step
(a launch repeated back to back re-reads them from the 256-MB Infinity Cache).  Reference point only: the product
path never calls the library.  Run on the GPU box:  python tools/blas_table.py [--ms 10917,11143,37748]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mvp_pytorch_amd import hip  # noqa: E402

dev = torch.device("cuda:0")


def rnd(*s):
    return (torch.randn(*s, device=dev) * 0.5).to(torch.bfloat16)


def cold_us(fn, flush, reps=5):
    fn()
    torch.cuda.synchronize()
    tot = 0.0
    for r in range(reps):
        flush.fill_(r)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        tot += e0.elapsed_time(e1)
    return tot / reps * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ms", default="10917,11143,37748,19200,64000")
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--ab", action="store_true", help="diagnostic library (MVPTR_LIB=diag): also time the persistent ring "
                                                      "experiment (MVPTR_GEMM_CFG=p, gemm_ntp_kernel)")
    ap.add_argument("--cfg", default="p", help="with --ab: the MVPTR_GEMM_CFG value of the extra column (p, v4, w4, s128, t256k, 8)")
    ap.add_argument("--loop-only", action="store_true", help="with --ab: two more columns, both kernels without their epilogue (MVPTR_NT_EXP bit 10)")
    args = ap.parse_args()
    H, I = 768, 3072
    base_exp = int(os.environ.get("MVPTR_NT_EXP", "0") or 0)      # the run's own experiment flags survive the loop-only columns
    flush = torch.empty(768 << 20, dtype=torch.uint8, device=dev)
    print("%-22s %6s %5s %5s  %9s %8s  %9s %8s  %6s%s" % ("gemm (epilogue)", "M", "N", "K", "ours us", "TF/s", "blasLt us", "TF/s", "ratio",
                                                           ("   %s us" % args.cfg) if args.ab else ""))
    tot_ours = tot_lib = tot_old = 0.0
    for M in [int(v) for v in args.ms.split(",")]:
        x, xi, x3 = rnd(M, H), rnd(M, I), rnd(M, 3 * H)
        shapes = [("qkv fwd BIAS", x, rnd(3 * H, H), hip.EPI_BIAS, None),
                  ("attn-out fwd RESID", x, rnd(H, H), hip.EPI_BIAS_RESID, x),
                  ("ffn1 fwd GELU", x, rnd(I, H), hip.EPI_BIAS_GELU, None),
                  ("ffn2 fwd RESID", xi, rnd(H, I), hip.EPI_BIAS_RESID, x),
                  ("ffn2 dgrad GELU_BWD", x, rnd(I, H), hip.EPI_GELU_BWD, torch.randint(0, 256, (M, I), device=dev, dtype=torch.uint8)),
                  ("ffn1 dgrad ADD", xi, rnd(H, I), hip.EPI_ADD, x),
                  ("attn-out dgrad ADD", x, rnd(H, H), hip.EPI_ADD, None),
                  ("qkv dgrad ADD", x3, rnd(H, 3 * H), hip.EPI_ADD, x)]
        for name, a, b, epi, aux in shapes:
            N, K = b.shape
            bias = torch.zeros(N, device=dev)
            out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
            out0 = torch.empty(M, N, device=dev, dtype=torch.uint8) if epi == hip.EPI_BIAS_GELU else out   # the 8-bit gelu' stash
            out1 = torch.empty(M, N, device=dev, dtype=torch.bfloat16) if epi == hip.EPI_BIAS_GELU else None
            vec = torch.zeros(N, device=dev) if epi == hip.EPI_GELU_BWD else None
            ours = cold_us(lambda: hip.gemm_nt(a, b, epi, bias=bias, aux=aux, out=out0, out1=out1, vec_out=vec), flush, args.reps)
            old = None
            loop_txt = ""
            if args.ab:
                hip.set_knob("MVPTR_GEMM_CFG", args.cfg)
                old = cold_us(lambda: hip.gemm_nt(a, b, epi, bias=bias, aux=aux, out=out0, out1=out1, vec_out=vec), flush, args.reps)
                hip.set_knob("MVPTR_GEMM_CFG", "")
                tot_old += old
                if args.loop_only:
                    hip.set_knob("MVPTR_NT_EXP", str(base_exp | 1024))
                    l0 = cold_us(lambda: hip.gemm_nt(a, b, epi, bias=bias, aux=aux, out=out0, out1=out1, vec_out=vec), flush, args.reps)
                    hip.set_knob("MVPTR_GEMM_CFG", args.cfg)
                    l1 = cold_us(lambda: hip.gemm_nt(a, b, epi, bias=bias, aux=aux, out=out0, out1=out1, vec_out=vec), flush, args.reps)
                    hip.set_knob("MVPTR_GEMM_CFG", "")
                    hip.set_knob("MVPTR_NT_EXP", str(base_exp))
                    loop_txt = "   loop-only %7.1f / %7.1f" % (l0, l1)
            bt = b.t()
            lib = cold_us(lambda: torch.matmul(a, bt, out=out), flush, args.reps)
            fl = 2.0 * M * N * K
            tot_ours += ours
            tot_lib += lib
            print("%-22s %6d %5d %5d  %9.1f %8.1f  %9.1f %8.1f  %6.2f%s" % (name, M, N, K, ours, fl / ours / 1e6, lib, fl / lib / 1e6, ours / lib,
                                                                             ("   %8.1f" % old if old is not None else "") + loop_txt))
    print("sum: ours %.1f us, hipBLASLt (no epilogue) %.1f us%s" % (tot_ours, tot_lib, ", %s %.1f us" % (args.cfg, tot_old) if args.ab else ""))


if __name__ == "__main__":
    main()
