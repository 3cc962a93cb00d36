#!/usr/bin/env python
"""Launch sequence of the LAST step in a rocprofv3 kernel trace (bench.py --one-stream): every kernel in start order with its
duration and the gap in front of it; runs of the big GEMM / attention / LayerNorm kernels are folded into one line.  Shows the
chains of small kernels that sit between the encoder stacks.   step_sequence.py TRACE.csv"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last step starts at the last arena zero fill (the biggest FillFunctor<float>)
fills = [i for i, r in enumerate(rows) if "FillFunctor<float>" in r["Kernel_Name"] and int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) > 60000]
bounds = fills + [len(rows)]
steps = [(bounds[i], bounds[i + 1]) for i in range(len(fills)) if bounds[i + 1] - bounds[i] > 300]      # a step = fill ... next fill
start, stop = steps[-1] if steps else (0, len(rows))
big = ("gemm_nt8", "gemm_tn_sk", "attn_", "ln_fwd_j", "ln_bwd_j", "ln_bwd_finalize")
prev_end = int(rows[start]["Start_Timestamp"])
t0 = prev_end
run_n, run_t, small_t, small_n = 0, 0.0, 0.0, 0
for r in rows[start:stop]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    d, gap = (e - s) / 1e3, (s - prev_end) / 1e3
    name = r["Kernel_Name"]
    if any(b in name for b in big):
        run_n += 1
        run_t += d
    else:
        if run_n:
            print("        ... %d stack kernels, %.1f us" % (run_n, run_t))
            run_n, run_t = 0, 0.0
        small_t += d
        small_n += 1
        print("%9.1f us  +gap %6.1f  dur %7.1f  %s" % ((s - t0) / 1e3, gap, d, name[:110]))
    prev_end = max(prev_end, e)
if run_n:
    print("        ... %d stack kernels, %.1f us" % (run_n, run_t))
print("small kernels of the step: %d launches, %.1f us" % (small_n, small_t))
