#!/usr/bin/env python
"""Known-byte-count streaming reads for calibrating rocprofv3's FETCH_SIZE on gfx950:
    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d DIR -- python3 tools/calib_fetch.py
reads a 1-GiB buffer (far beyond the 256-MiB Infinity Cache) once per launch through the LDS-DMA
path (buffer_load_dwordx4 ... lds, the GEMM operand path) and through global_load_dwordx4; the
counter value per launch of stream_read_kernel divided by 2^30 is the calibration factor
(tools/traffic_json.py applies 1/factor to FETCH_SIZE of the GEMM kernels)."""
import os
import sys

import torch

os.environ.setdefault("MVPTR_LIB", "diag")   # the probes are exported by the diagnostic build only (include/mvptr_diag.h)

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mvp_pytorch_amd import hip  # noqa: E402

dev = torch.device("cuda:0")
buf = torch.empty(1 << 30, dtype=torch.uint8, device=dev).random_(0, 255)
other = torch.empty(1 << 30, dtype=torch.uint8, device=dev)
sink = torch.zeros(1, device=dev)
for mode in (0, 1, 0, 1):
    other.random_(0, 255)   # evict the buffer from the Infinity Cache between launches
    hip.diag_stream_read(buf, mode, sink)
torch.cuda.synchronize()
print("calib done: 2 launches per mode of %d bytes (grid order: mode 0, 1, 0, 1)" % buf.numel())
