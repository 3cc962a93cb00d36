#!/usr/bin/env python
"""Diagnostic: wall-clock timeline of every gemm_nt workgroup (s_memrealtime at start / end of the
MFMA loop / after the last store has been acknowledged) from the -DMVPTR_TIMELINE_BUILD library.
Prints, per 2-us bin, how many workgroups are in their loop and how many in their epilogue, and the
per-workgroup phase durations.  Answers: do the workgroups run their epilogues (HBM bursts) in
lockstep?"""
import os
os.environ.setdefault("MVPTR_LIB", "diag")   # kernel-configuration knobs live in the diagnostic build only (make -C mvp_pytorch_amd/csrc diag)
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mvp_pytorch_amd import hip  # noqa: E402

hip.LIB_PATH = os.path.join(ROOT, "mvp_pytorch_amd", "csrc", "libmvptr_hip_tl.so")
dev = torch.device("cuda:0")


def run(cfg, M, N, K, epi, name, bm, bn):
    hip.set_knob("MVPTR_GEMM_CFG", cfg)
    a = (torch.randn(M, K, device=dev) * 0.5).to(torch.bfloat16)
    b = (torch.randn(N, K, device=dev) * 0.5).to(torch.bfloat16)
    bias = torch.zeros(N, device=dev)
    aux = torch.randn(M, N, device=dev).to(torch.bfloat16) if epi == hip.EPI_BIAS_RESID else None
    if epi == hip.EPI_GELU_BWD:
        aux = torch.randint(0, 256, (M, N), device=dev, dtype=torch.uint8)
    nwg = ((M + bm - 1) // bm) * ((N + bn - 1) // bn)
    st = torch.zeros(nwg * 8, dtype=torch.int64, device=dev)
    hip.set_knob("MVPTR_GEMM_STAMPS", str(st.data_ptr()))
    out = torch.empty(M, N, device=dev, dtype=torch.uint8 if epi == hip.EPI_BIAS_GELU else torch.bfloat16)
    out1 = torch.empty(M, N, device=dev, dtype=torch.bfloat16) if epi == hip.EPI_BIAS_GELU else None
    vec = torch.zeros(N, device=dev) if epi == hip.EPI_GELU_BWD else None
    for _ in range(3):
        hip.gemm_nt(a, b, epi, bias=bias, aux=aux, out=out, out1=out1, vec_out=vec)
    torch.cuda.synchronize()
    s = st.view(nwg, 8).cpu().numpy()
    t0 = s[:, 0].min()
    start, loop, end = (s[:, 0] - t0) / 100.0, (s[:, 1] - t0) / 100.0, (s[:, 2] - t0) / 100.0   # us
    cu = (s[:, 4] & 0xf) * 4096 + (s[:, 3] & 0xff00)  # xcc, se/sh/cu bits of HW_ID
    print("== %s cfg=%s M=%d N=%d K=%d: %d workgroups on %d distinct CUs, kernel span %.1f us" % (name, cfg, M, N, K, nwg, len(np.unique(cu)), end.max()))
    print("   loop duration us: mean %.1f  p10 %.1f  p90 %.1f | epilogue (to last store ack): mean %.1f  p10 %.1f  p90 %.1f"
          % ((loop - start).mean(), np.percentile(loop - start, 10), np.percentile(loop - start, 90),
             (end - loop).mean(), np.percentile(end - loop, 10), np.percentile(end - loop, 90)))
    order = np.argsort(start)
    rounds = np.array_split(order, max(1, int(round(nwg / len(np.unique(cu)) / (2 if cfg.startswith("w") else 1)))))
    for i, r in enumerate(rounds):
        print("   dispatch round %d: start %.1f +- %.1f us, loop %.1f, epilogue %.1f" % (i, start[r].mean(), start[r].std(), (loop - start)[r].mean(), (end - loop)[r].mean()))
    width = 4.0
    bins = np.arange(0, end.max() + width, width)
    line_l, line_e = [], []
    for b0 in bins:
        mid = b0 + width / 2
        line_l.append(int(((start <= mid) & (mid < loop)).sum()))
        line_e.append(int(((loop <= mid) & (mid < end)).sum()))
    print("   in loop     per %.0f-us bin: %s" % (width, " ".join("%3d" % x for x in line_l)))
    print("   in epilogue per %.0f-us bin: %s" % (width, " ".join("%3d" % x for x in line_e)))


run("w4", 32000, 3072, 768, hip.EPI_BIAS_GELU, "ffn1 fwd GELU", 256, 128)
run("t256k", 32000, 3072, 768, hip.EPI_BIAS_GELU, "ffn1 fwd GELU", 256, 256)
run("w4", 32000, 3072, 768, hip.EPI_GELU_BWD, "ffn2 dgrad GELU_BWD", 256, 128)
run("w4", 32000, 2304, 768, hip.EPI_BIAS, "qkv fwd BIAS", 256, 128)
run("w4", 32000, 768, 3072, hip.EPI_BIAS_RESID, "ffn2 fwd RESID", 256, 128)
