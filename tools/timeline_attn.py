#!/usr/bin/env python
"""Per-workgroup phase times of attn_bwd_kernel from the -DMVPTR_TIMELINE_BUILD library (s_memrealtime at entry,
after the staging barrier, when the last wave is done): where a (sequence, head) workgroup spends its time, by
sequence length, for the row-packed joint pass of the timed batch."""
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
torch.manual_seed(0)
B, heads, H = 512, 12, 768
lens = torch.randint(21, 126, (B,), dtype=torch.int32)
starts = (torch.cumsum(lens, 0, dtype=torch.int32) - lens)
rows, Lmax = int(lens.sum()), int(lens.max())
qkv = (torch.randn(rows, 3 * H, device=dev) * 0.5).to(torch.bfloat16)
dctx = (torch.randn(rows, H, device=dev) * 0.5).to(torch.bfloat16)
if len(sys.argv) > 1 and sys.argv[1] == "sorted":      # longest sequences first (what the encoder passes since round 2)
    order = torch.argsort(lens, descending=True)
    lens, starts = lens[order].contiguous(), starts[order].contiguous()
lens_d, starts_d = lens.to(dev), starts.to(dev)
drop = hip.make_dropout(0.1, 1234)
lib = hip.load()
ctx = torch.empty(rows, H, device=dev, dtype=torch.bfloat16)
lse = torch.empty(B, heads, Lmax, device=dev, dtype=torch.float32)
P = hip._p
hip._check(lib.mvptr_attention_fwd_packed(P(qkv), None, P(ctx), P(lse), P(starts_d), P(lens_d), B, Lmax, heads, hip._dp(drop), hip._stream()))
dq = torch.empty_like(qkv)
st = torch.zeros(B * heads * 8, dtype=torch.int64, device=dev)
hip.set_knob("MVPTR_GEMM_STAMPS", str(st.data_ptr()))


def bwd():
    hip._check(lib.mvptr_attention_bwd_packed(P(qkv), None, P(ctx), P(dctx), P(lse), P(dq), P(starts_d), P(lens_d), B, Lmax, heads, hip._dp(drop), hip._stream()))


for _ in range(3):
    bwd()
torch.cuda.synchronize()
st.zero_()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
bwd()
e1.record()
torch.cuda.synchronize()
s = st.view(-1, 8).cpu().numpy()
pro, loop, L = (s[:, 1] - s[:, 0]) / 100.0, (s[:, 2] - s[:, 1]) / 100.0, s[:, 3]
print("launch %.1f us for %d workgroups (%d rows); per workgroup: prologue (stage 4 tiles + delta + barrier) mean %.2f us, tasks + stores mean %.2f us"
      % (e0.elapsed_time(e1) * 1e3, len(s), rows, pro.mean(), loop.mean()))
for lo, hi in ((1, 32), (33, 64), (65, 96), (97, 128)):
    m = (L >= lo) & (L <= hi)
    if m.any():
        print("  L %3d-%3d (nb=%d): %5d workgroups, prologue %.2f us, tasks %.2f us" % (lo, hi, hi // 32, m.sum(), pro[m].mean(), loop[m].mean()))
if s[:, 4].any():    # the one-pass kernel also stamps the end of its phase 1 (P, dS, dK, dV), slowest wave
    p1 = (s[:, 4] - s[:, 1]) / 100.0
    for lo, hi in ((1, 32), (33, 64), (65, 96), (97, 128)):
        m = (L >= lo) & (L <= hi)
        if m.any():
            print("  L %3d-%3d: phase 1 %.2f us, dS hand-over + dK/dV stores + phase 2 (dQ) %.2f us" % (lo, hi, p1[m].mean(), (loop[m] - p1[m]).mean()))
span = (s[:, 2].max() - s[:, 0].min()) / 100.0
print("  kernel span from the stamps %.1f us; sum of workgroup times / 512 slots = %.1f us" % (span, (pro + loop).sum() / 512))
