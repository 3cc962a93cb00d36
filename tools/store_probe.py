#!/usr/bin/env python
"""Per-CU store rate by access shape (mvptr_diag_store_probe): 1 KiB per wave instruction written as
R row segments, rows `stride` bytes apart; 256 or 64 workgroups of 8 waves."""
import os
import sys

import torch

os.environ.setdefault("MVPTR_LIB", "diag")   # the probes are exported by the diagnostic build only (include/mvptr_diag.h)

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mvp_pytorch_amd import hip  # noqa: E402

dev = torch.device("cuda:0")
buf = torch.empty(3 << 30, dtype=torch.uint8, device=dev)


def timeit(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


for blocks in (256, 64, 8):
    for R, stride in ((1, 1024), (1, 4608), (2, 4608), (4, 4608), (8, 4608), (8, 1536), (8, 6144), (16, 4608), (32, 4608), (64, 4608)):
        bpw = 128 * 1024   # bytes per wave
        need = blocks * 8 * (bpw // 1024) * R * stride
        if need > buf.numel():
            continue
        us = timeit(lambda: hip.diag_store_probe(buf, blocks, bpw, R, stride))
        tot = blocks * 8 * bpw
        print("blocks %3d  %2d rows x %4d B, stride %5d:  %7.1f us  %6.2f TB/s  %6.1f GB/s per workgroup" %
              (blocks, R, 1024 // R, stride, us, tot / us / 1e6, tot / blocks / us / 1e3), flush=True)
