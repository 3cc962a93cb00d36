#!/usr/bin/env python
"""Where does a GEMM epilogue's time go?  (FFN1 + GELU 246 us = 143 loop + 103 epilogue, GELU backward 234 = 131 + 102.)
Diagnostic build only (MVPTR_NT_EXP bits 26-30 = GemmNtArgs.epi_ablate): the same cold launch with parts of the epilogue switched
off — 1 aux rows not loaded, 2 outputs not stored, 4 no column-sum atomics, 8 GELU arithmetic skipped, 16 the A&S GELU of rounds
1-5 — beside the loop-only launch (bit 10).  Also times the region-feature cast (mvptr_cast_pack) at the configs[1] shape.
Run on the GPU box:  python tools/epi_ablate.py [--ms 37748,10917]"""
import argparse
import os
import sys

import torch

os.environ.setdefault("MVPTR_LIB", "diag")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mvp_pytorch_amd import hip  # noqa: E402

dev = torch.device("cuda:0")


def rnd(*s):
    return (torch.randn(*s, device=dev) * 0.5).to(torch.bfloat16)


def cold_us(fn, flush, reps=7):
    fn()
    torch.cuda.synchronize()
    ts = []
    for r in range(reps):
        flush.fill_(r)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort()
    return sum(ts[1:-1]) / (len(ts) - 2)      # trimmed mean


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ms", default="37748,10917")
    args = ap.parse_args()
    H, I = 768, 3072
    flush = torch.empty(768 << 20, dtype=torch.uint8, device=dev)
    variants = [("full", 0), ("loop only", -1), ("A&S gelu (r1-5)", 16), ("no gelu math", 8), ("no stores", 2), ("no aux loads", 1),
                ("no colsum", 4), ("no aux, no colsum", 5), ("no stores, no aux", 3), ("no st/aux/colsum/math", 15),
                # cache policy of the 8-bit stash stores (MVPTR_NT_EXP bit 9 / bits 19-21; product: non-temporal)
                ("stash plain", -512), ("stash sc1", -(1 << 19)), ("stash sc0sc1", -(2 << 19)), ("stash nt buf", -(3 << 19))]
    for M in [int(v) for v in args.ms.split(",")]:
        x, xi, x3 = rnd(M, H), rnd(M, I), rnd(M, 3 * H)
        shapes = [("qkv fwd BIAS", x, rnd(3 * H, H), hip.EPI_BIAS, None),
                  ("attn-out fwd RESID", x, rnd(H, H), hip.EPI_BIAS_RESID, x),
                  ("ffn1 fwd GELU", x, rnd(I, H), hip.EPI_BIAS_GELU, None),
                  ("ffn1 fwd GELU bf16 stash", x, rnd(I, H), hip.EPI_BIAS_GELU_BF16, None),
                  ("ffn2 fwd RESID", xi, rnd(H, I), hip.EPI_BIAS_RESID, x),
                  ("ffn2 dgrad GELU_BWD", x, rnd(I, H), hip.EPI_GELU_BWD, torch.randint(0, 256, (M, I), device=dev, dtype=torch.uint8)),
                  ("ffn2 dgrad GELU_BWD bf16", x, rnd(I, H), hip.EPI_GELU_BWD_BF16, rnd(M, I)),
                  ("ffn2 dgrad GELU_BWD no vec", x, rnd(I, H), hip.EPI_GELU_BWD, torch.randint(0, 256, (M, I), device=dev, dtype=torch.uint8)),
                  ("ffn2 dgrad GELU_BWD16 no vec", x, rnd(I, H), hip.EPI_GELU_BWD_BF16, rnd(M, I)),
                  ("ffn1 dgrad ADD", xi, rnd(H, I), hip.EPI_ADD, x),
                  ("qkv dgrad ADD", x3, rnd(H, 3 * H), hip.EPI_ADD, x)]
        print("M = %d, cold us" % M)
        print("%-26s" % "gemm" + "".join("%22s" % v[0] for v in variants))
        for name, a, b, epi, aux in shapes:
            N, K = b.shape
            bias = torch.zeros(N, device=dev)
            out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
            two = epi in (hip.EPI_BIAS_GELU, hip.EPI_BIAS_GELU_BF16)
            out0 = torch.empty(M, N, device=dev, dtype=torch.uint8) if epi == hip.EPI_BIAS_GELU else out
            if epi == hip.EPI_BIAS_GELU_BF16:
                out0 = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
            out1 = torch.empty(M, N, device=dev, dtype=torch.bfloat16) if two else None
            vec = torch.zeros(N, device=dev) if (epi in (hip.EPI_GELU_BWD, hip.EPI_GELU_BWD_BF16) and "no vec" not in name) else None
            row = "%-26s" % name
            for vname, bits in variants:
                exp = 1024 if bits == -1 else (-bits if bits < 0 else (bits << 26))
                hip.set_knob("MVPTR_NT_EXP", str(exp))
                us = cold_us(lambda: hip.gemm_nt(a, b, epi, bias=bias, aux=aux, out=out0, out1=out1, vec_out=vec), flush)
                row += "%22.1f" % us
            hip.set_knob("MVPTR_NT_EXP", "0")
            print(row, flush=True)
    # region features: 256 pairs x 50 regions x 2054 f32 -> K-padded bf16
    rows, D = 256 * 50, 2054
    feats = torch.randn(rows, D, device=dev)
    fb = torch.empty(rows, 2056, device=dev, dtype=torch.bfloat16)
    us = cold_us(lambda: hip.cast_pack(feats, dst=fb), flush, reps=9)
    nbytes = rows * D * 4 + rows * 2056 * 2
    print("region-feature cast %d x %d f32 -> bf16: %.1f us = %.2f TB/s (read + write)" % (rows, D, us, nbytes / us / 1e6))
    ref = feats.to(torch.bfloat16)
    assert torch.equal(fb[:, :D], ref) and (fb[:, D:] == 0).all()
    us = cold_us(lambda: feats.to(torch.bfloat16), flush, reps=9)
    print("  torch .to(bfloat16) of the same tensor (no K padding): %.1f us" % us)
    w = rnd(768, 2056)
    us = cold_us(lambda: hip.gemm_nt(fb, w, hip.EPI_BIAS, bias=torch.zeros(768, device=dev)), flush, reps=9)
    print("img-embedding GEMM 12800 x 768 x 2056: %.1f us = %.0f TFLOP/s" % (us, 2.0 * rows * 768 * 2056 / us / 1e6))


if __name__ == "__main__":
    main()
