#!/usr/bin/env python
"""In-kernel clock and cycles per 32-row stage of the gemm_tn "H" loop (diagnostic -DMVPTR_TIMELINE_BUILD
library: s_memtime / s_memrealtime around the loop, after >= 2 s of back-to-back launches on random
data; MI355X_MICROARCH.md DVFS give-back item 6).  One stage = 64 v_mfma_f32_16x16x32_bf16 per wave =
1024 MFMA cycles."""
import os
os.environ.setdefault("MVPTR_LIB", "diag")   # kernel-configuration knobs live in the diagnostic build only (make -C mvp_pytorch_amd/csrc diag)
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mvp_pytorch_amd import hip  # noqa: E402

hip.LIB_PATH = os.path.join(ROOT, "mvp_pytorch_amd", "csrc", os.environ.get("MVPTR_TOOL_LIB", "libmvptr_hip_tl.so"))
dev = torch.device("cuda:0")
CFG = sys.argv[1] if len(sys.argv) > 1 else "h"
for M in (64000, 10917):
    shapes = [(768, 3072), (3072, 768)]
    probs = []
    for N, K in shapes:
        probs.append(((torch.randn(M, N, device=dev) * 0.5).to(torch.bfloat16), (torch.randn(M, K, device=dev) * 0.5).to(torch.bfloat16),
                      torch.zeros(N, K, device=dev), None))
    st = torch.zeros(4096 * 8, dtype=torch.int64, device=dev)
    hip.set_knob("MVPTR_GEMM_TN", CFG)
    hip.set_knob("MVPTR_GEMM_STAMPS", str(st.data_ptr()))
    t0 = time.time()
    n = 0
    while time.time() - t0 < 2.5:
        for _ in range(20):
            hip.gemm_tn_multi(probs)
        torch.cuda.synchronize()
        n += 20
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        hip.gemm_tn_multi(probs)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 100.0
    s = st.view(-1, 8).cpu().numpy()
    s = s[s[:, 4] > 0]
    cyc, ticks, steps = (s[:, 3] - s[:, 2]).astype(np.float64), (s[:, 1] - s[:, 0]).astype(np.float64), s[:, 4].astype(np.float64)
    clk = cyc / ticks * 0.1          # GHz
    print("M=%d cfg=%s: %.1f us per launch; %d workgroups; stages per workgroup %.0f; loop %.1f us (median); "
          "cycles per stage %.0f (median; 1024 = MFMA-bound); in-loop clock %.2f GHz (median, p10 %.2f, p90 %.2f)"
          % (M, CFG, us, len(s), np.median(steps), np.median(ticks) / 100.0, np.median(cyc / steps), np.median(clk),
             np.percentile(clk, 10), np.percentile(clk, 90)))
