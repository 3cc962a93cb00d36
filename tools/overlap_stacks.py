#!/usr/bin/env python
"""Experiment: the text and the visual stack are independent; at the packed row counts (~11 k rows)
their N = 768 GEMMs give a 256x256 tile to only half of the CUs.  One stream vs two streams."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mvp_pytorch_amd import hip  # noqa: E402

dev = torch.device("cuda:0")
H, I = 768, 3072


def rnd(*s):
    return (torch.randn(*s, device=dev) * 0.5).to(torch.bfloat16)


class Stack:
    def __init__(self, M):
        self.M = M
        self.x, self.xi = rnd(M, H), rnd(M, I)
        self.wq, self.wo, self.wi, self.wout = rnd(3 * H, H), rnd(H, H), rnd(I, H), rnd(H, I)
        self.b3, self.bh, self.bi = torch.zeros(3 * H, device=dev), torch.zeros(H, device=dev), torch.zeros(I, device=dev)
        self.o3, self.oh = torch.empty(M, 3 * H, device=dev, dtype=torch.bfloat16), torch.empty(M, H, device=dev, dtype=torch.bfloat16)
        self.oi, self.oi2 = torch.empty(M, I, device=dev, dtype=torch.uint8), torch.empty(M, I, device=dev, dtype=torch.bfloat16)
        self.g = torch.ones(H, device=dev)

    def layer(self):
        hip.gemm_nt(self.x, self.wq, hip.EPI_BIAS, bias=self.b3, out=self.o3)
        hip.gemm_nt(self.x, self.wo, hip.EPI_BIAS_RESID, bias=self.bh, aux=self.x, out=self.oh)
        hip.layernorm_fwd(self.oh, self.g, self.bh, 1e-12, save_stats=False)
        hip.gemm_nt(self.x, self.wi, hip.EPI_BIAS_GELU, bias=self.bi, out=self.oi, out1=self.oi2)
        hip.gemm_nt(self.xi, self.wout, hip.EPI_BIAS_RESID, bias=self.bh, aux=self.x, out=self.oh)
        hip.layernorm_fwd(self.oh, self.g, self.bh, 1e-12, save_stats=False)


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


a, b = Stack(10917), Stack(11143)
side, main = torch.cuda.Stream(), torch.cuda.current_stream()


def seq():
    for _ in range(6):
        a.layer()
    for _ in range(6):
        b.layer()


def par():
    side.wait_stream(main)
    with torch.cuda.stream(side):
        for _ in range(6):
            b.layer()
    for _ in range(6):
        a.layer()
    main.wait_stream(side)


print("6+6 forward layers at M = 10917 / 11143: one stream %.0f us, two streams %.0f us" % (bench(seq), bench(par)))

# proxy for ONE grouped launch per GEMM over both stacks: a single stack with the rows of both (same weights — the
# grouped kernel would read two weight sets of the same size, so tile counts, rounds and bytes are those of this proxy)
c = Stack(10917 + 11143)


def merged():
    for _ in range(6):
        c.layer()


t_seq, t_par, t_mrg = bench(seq), bench(par), bench(merged)
print("proxy for grouped launches over both stacks (one stack of %d rows, 6 forward layers): %.0f us  (one stream %.0f, two streams %.0f)"
      % (c.M, t_mrg, t_seq, t_par))
