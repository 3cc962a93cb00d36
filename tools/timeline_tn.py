#!/usr/bin/env python
"""Diagnostic: wall-clock timeline of the gemm_tn workgroups (loop vs atomic write-out), from the
-DMVPTR_TIMELINE_BUILD library."""
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


def run(M, N, K, name):
    dy = (torch.randn(M, N, device=dev) * 0.5).to(torch.bfloat16)
    x = (torch.randn(M, K, device=dev) * 0.5).to(torch.bfloat16)
    dw = torch.zeros(N, K, device=dev)
    st = torch.zeros(8192 * 8, dtype=torch.int64, device=dev)
    hip.set_knob("MVPTR_GEMM_STAMPS", str(st.data_ptr()))
    for _ in range(3):
        st.zero_()
        hip.gemm_tn(dy, x, dw)
    torch.cuda.synchronize()
    s = st.view(-1, 8).cpu().numpy()
    s = s[s[:, 0] > 0]
    t0 = s[:, 0].min()
    start, loop, end = (s[:, 0] - t0) / 100.0, (s[:, 1] - t0) / 100.0, (s[:, 2] - t0) / 100.0
    print("== %s M=%d N=%d K=%d: %d workgroups, span %.1f us (%.0f TF); loop %.1f us (p10 %.1f p90 %.1f), write-out %.1f us (p10 %.1f p90 %.1f); starts: %.1f..%.1f"
          % (name, M, N, K, len(s), end.max(), 2.0 * M * N * K / end.max() / 1e6, (loop - start).mean(), np.percentile(loop - start, 10),
             np.percentile(loop - start, 90), (end - loop).mean(), np.percentile(end - loop, 10), np.percentile(end - loop, 90), start.min(), start.max()))
    width = 10.0
    bins = np.arange(0, end.max() + width, width)
    print("   in loop      per %.0f-us bin: %s" % (width, " ".join("%3d" % int(((start <= b + width / 2) & (b + width / 2 < loop)).sum()) for b in bins)))
    print("   in write-out per %.0f-us bin: %s" % (width, " ".join("%3d" % int(((loop <= b + width / 2) & (b + width / 2 < end)).sum()) for b in bins)))


for cfg in ("32", "K"):
    if cfg is None:
        hip.set_knob("MVPTR_GEMM_TN", "")
    else:
        hip.set_knob("MVPTR_GEMM_TN", cfg)
    print("MVPTR_GEMM_TN =", cfg)
    run(64000, 3072, 768, "w_i")
    run(64000, 768, 768, "w_o")
    run(19200, 2304, 768, "w_qkv")
