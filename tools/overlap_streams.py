#!/usr/bin/env python
"""Experiment: run the weight-gradient GEMM (gemm_tn) on a second stream beside the data-gradient
GEMM (gemm_nt) of the same layer.  Sequential time vs two-stream time."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mvp_pytorch_amd import hip  # noqa: E402

dev = torch.device("cuda:0")
H, I = 768, 3072


def rnd(*s):
    return (torch.randn(*s, device=dev) * 0.5).to(torch.bfloat16)


def bench(fn, reps=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


for M in (64000, 19200):
    x, xi, x3 = rnd(M, H), rnd(M, I), rnd(M, 3 * H)
    w_it, w_outt, w_qkvt, w_ot = rnd(H, I), rnd(I, H), rnd(H, 3 * H), rnd(H, H)
    dwi, dwo, dwq, dwa = (torch.zeros(I, H, device=dev), torch.zeros(H, I, device=dev),
                          torch.zeros(3 * H, H, device=dev), torch.zeros(H, H, device=dev))
    o_i, o_h, vec = torch.empty(M, I, device=dev, dtype=torch.bfloat16), torch.empty(M, H, device=dev, dtype=torch.bfloat16), torch.zeros(I, device=dev)
    side = torch.cuda.Stream(priority=0)
    main = torch.cuda.current_stream()

    # the GEMM part of one layer's backward: [dgrad ; wgrad] x 4
    def dgrads():
        hip.gemm_nt(x, w_outt, hip.EPI_GELU_BWD, aux=xi, out=o_i, vec_out=vec)      # d2 W_out .* g'
        hip.gemm_nt(xi, w_it, hip.EPI_ADD, aux=x, out=o_h)                          # dU W_i + resid
        hip.gemm_nt(x, w_ot, hip.EPI_ADD, out=o_h)                                  # attn-out dgrad
        hip.gemm_nt(x3, w_qkvt, hip.EPI_ADD, aux=x, out=o_h)                        # qkv dgrad

    def wgrads():
        hip.gemm_tn(x, xi, dwo)     # dW_out = d2^T a
        hip.gemm_tn(xi, x, dwi)     # dW_i = dU^T x1
        hip.gemm_tn(x, x, dwa)      # dW_o
        hip.gemm_tn(x3, x, dwq)     # dW_qkv

    def seq():
        dgrads()
        wgrads()

    def par():
        ev = torch.cuda.Event()
        ev.record(main)
        with torch.cuda.stream(side):
            side.wait_event(ev)
            wgrads()
            ev2 = torch.cuda.Event()
            ev2.record(side)
        dgrads()
        main.wait_event(ev2)

    def inter():  # interleaved issue order, two streams
        ev = torch.cuda.Event()
        ev.record(main)
        side.wait_event(ev)
        hip.gemm_nt(x, w_outt, hip.EPI_GELU_BWD, aux=xi, out=o_i, vec_out=vec)
        with torch.cuda.stream(side):
            hip.gemm_tn(x, xi, dwo)
        hip.gemm_nt(xi, w_it, hip.EPI_ADD, aux=x, out=o_h)
        with torch.cuda.stream(side):
            hip.gemm_tn(xi, x, dwi)
        hip.gemm_nt(x, w_ot, hip.EPI_ADD, out=o_h)
        with torch.cuda.stream(side):
            hip.gemm_tn(x, x, dwa)
        hip.gemm_nt(x3, w_qkvt, hip.EPI_ADD, aux=x, out=o_h)
        with torch.cuda.stream(side):
            hip.gemm_tn(x3, x, dwq)
            ev2 = torch.cuda.Event()
            ev2.record(side)
        main.wait_event(ev2)

    td, tw = bench(dgrads), bench(wgrads)
    print("M=%d: dgrads alone %.0f us, wgrads alone %.0f us, sum %.0f | one stream %.0f | two streams %.0f | two streams interleaved issue %.0f"
          % (M, td, tw, td + tw, bench(seq), bench(par), bench(inter)), flush=True)
