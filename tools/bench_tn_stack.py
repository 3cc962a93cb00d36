#!/usr/bin/env python
"""Round 5: every weight gradient of an encoder stack in ONE balanced launch (mvptr_gemm_tn_stack) against the per-layer
grouped launches (mvptr_gemm_tn_multi_ws: FFN pair + attention pair per layer) on the operand sets of a configs[1] step.
Checks the results against each other and times both (operands of six layers are > 256 MB: they come from HBM as in the step).
Run on the GPU box:  python tools/bench_tn_stack.py [--ms 10917,37748] [--layers 6]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mvp_pytorch_amd import hip  # noqa: E402

dev = torch.device("cuda:0")


def rnd(*s):
    return (torch.randn(*s, device=dev) * 0.5).to(torch.bfloat16)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ms", default="10917,37748,19200")
    ap.add_argument("--layers", type=int, default=6)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--H", type=int, default=768)
    ap.add_argument("--I", type=int, default=3072)
    ap.add_argument("--no-colsum", action="store_true", help="no bias-gradient column sums on FFN1 / Q|K|V (A/B of their cost)")
    args = ap.parse_args()
    H, I, L = args.H, args.I, args.layers
    for M in [int(v) for v in args.ms.split(",")]:
        layers = []
        for _ in range(L):
            d2, a, dU, x1, d1, ctx, dq, x = rnd(M, H), rnd(M, I), rnd(M, I), rnd(M, H), rnd(M, H), rnd(M, H), rnd(M, 3 * H), rnd(M, H)
            cs = not args.no_colsum
            layers.append([(d2, a, (H, I), False), (dU, x1, (I, H), cs), (d1, ctx, (H, H), False), (dq, x, (3 * H, H), cs)])

        def grads():
            return [[(torch.zeros(n, k, device=dev), torch.zeros(n, device=dev) if cs else None) for (_, _, (n, k), cs) in lay] for lay in layers]

        def per_layer(g):
            for lay, gl in zip(layers, g):
                hip.gemm_tn_multi([(lay[0][0], lay[0][1], gl[0][0], gl[0][1]), (lay[1][0], lay[1][1], gl[1][0], gl[1][1])])
                hip.gemm_tn_multi([(lay[3][0], lay[3][1], gl[3][0], gl[3][1]), (lay[2][0], lay[2][1], gl[2][0], gl[2][1])])

        def stack(g, rows_dev=None):
            probs = []
            for lay, gl in zip(layers, g):
                for (dy, xx, _, _), (dw, cs) in zip(lay, gl):
                    probs.append((dy, xx, dw, cs))
            hip.gemm_tn_stack(probs, rows_dev=rows_dev)

        g0, g1 = grads(), grads()
        per_layer(g0)
        stack(g1)
        torch.cuda.synchronize()
        worst = 0.0
        for la, lb in zip(g0, g1):
            for (wa, ca), (wb, cb) in zip(la, lb):
                worst = max(worst, float((wa - wb).abs().max() / wa.abs().max()))
                if ca is not None:
                    worst = max(worst, float((ca - cb).abs().max() / ca.abs().max()))
        # device-side row count: the first Mv rows only
        Mv = M - 777
        rd = torch.tensor([Mv], device=dev, dtype=torch.int32)
        g2 = grads()
        stack(g2, rows_dev=rd)
        ref = layers[0][1][0][:Mv].float().t() @ layers[0][1][1][:Mv].float()
        err_rd = float((g2[0][1][0] - ref).abs().max() / ref.abs().max())
        ref = layers[0][1][0].float().t() @ layers[0][1][1].float()
        err_f = float((g1[0][1][0] - ref).abs().max() / ref.abs().max())

        def time(fn):
            g = grads()
            fn(g)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(args.reps):
                fn(g)
            e1.record()
            torch.cuda.synchronize()
            return e0.elapsed_time(e1) / args.reps * 1e3

        t_layer, t_stack = time(per_layer), time(stack)
        flops = 2.0 * M * L * (2 * H * I + 4 * H * H)
        print("M=%6d layers=%d  per-layer %8.1f us (%6.1f TF/s)   stack %8.1f us (%6.1f TF/s)   ratio %.3f   max rel diff %.2e  vs f32 %.2e  rows_dev %.2e"
              % (M, L, t_layer, flops / t_layer * 1e-6, t_stack, flops / t_stack * 1e-6, t_stack / t_layer, worst, err_f, err_rd), flush=True)
        del layers, g0, g1, g2
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
