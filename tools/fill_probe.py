#!/usr/bin/env python
"""Operand-fill rate of the GEMM staging pattern alone (mvptr_diag_fill_probe): 256 workgroups, one
per CU, each streaming a private (or one shared) region in 32-KiB stages, three in flight.  Working
sets from L2-resident to HBM; LDS-DMA against loads to registers."""
import os
import sys

import torch

os.environ.setdefault("MVPTR_LIB", "diag")   # the probes are exported by the diagnostic build only (include/mvptr_diag.h)

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mvp_pytorch_amd import hip  # noqa: E402

dev = torch.device("cuda:0")
buf = torch.empty(2 << 30, dtype=torch.uint8, device=dev)
buf.random_(0, 255)
sink = torch.zeros(4, device=dev)
BLOCKS = int(sys.argv[1]) if len(sys.argv) > 1 else 256


def run(wg_bytes, shared, mode):
    total = 4 << 30                       # bytes streamed per measurement, all workgroups
    reps = max(1, total // (wg_bytes * BLOCKS))
    hip.diag_fill_probe(buf, BLOCKS, wg_bytes, reps, shared, mode, sink)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(3):
        e0.record()
        hip.diag_fill_probe(buf, BLOCKS, wg_bytes, reps, shared, mode, sink)
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1))
    moved = float(wg_bytes) * BLOCKS * reps
    return moved / best / 1e9, moved / best / 1e6 / BLOCKS     # TB/s chip, GB/s per workgroup


print("blocks %d; working set = wg_bytes x blocks (private) or wg_bytes (shared)" % BLOCKS)
for wg_kib in (32, 64, 128, 512, 2048, 8192):
    wg = wg_kib << 10
    if wg * BLOCKS > buf.numel():
        continue
    line = "wg %5d KiB (set %6.1f MiB)" % (wg_kib, wg * BLOCKS / 2 ** 20)
    for mode, name in ((0, "lds-dma"), (1, "to-regs")):
        tb, gb = run(wg, 0, mode)
        line += "  %s %5.2f TB/s %6.1f GB/s/CU" % (name, tb, gb)
    print(line, flush=True)
for wg_kib in (256, 1024, 4096):
    wg = wg_kib << 10
    line = "shared %5d KiB             " % wg_kib
    for mode, name in ((0, "lds-dma"), (1, "to-regs")):
        tb, gb = run(wg, 1, mode)
        line += "  %s %5.2f TB/s %6.1f GB/s/CU" % (name, tb, gb)
    print(line, flush=True)
