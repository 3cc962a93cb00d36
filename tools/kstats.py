#!/usr/bin/env python
"""Per-step kernel table from a rocprofv3 kernel_stats.csv of a bench.py run:  kstats.py STATS.csv STEPS [TOP]
(STEPS = warmup + timed steps of the run; launches of the first steps' one-time work are averaged in)."""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
steps = float(sys.argv[2])
top = int(sys.argv[3]) if len(sys.argv) > 3 else 40
ours = ("anonymous namespace", "_GLOBAL__N_")
tab, tot, n, lib_t, lib_n = [], 0.0, 0.0, 0.0, 0.0
for r in rows:
    c, t = int(r["Calls"]), float(r["TotalDurationNs"]) / 1e6
    mine = any(o in r["Name"] for o in ours) and "at::native" not in r["Name"]
    tab.append((t / steps, c / steps, float(r["AverageNs"]) / 1e3, ("  " if mine else "T ") + r["Name"][:110]))
    tot += t
    n += c
    if not mine:
        lib_t += t
        lib_n += c
print("kernel time %.2f ms/step in %.1f launches/step; not from this library (torch / rocBLAS / runtime copies, marked T): "
      "%.2f ms in %.1f launches" % (tot / steps, n / steps, lib_t / steps, lib_n / steps))
for t, c, avg, nm in sorted(tab, reverse=True)[:top]:
    print("%7.3f ms  %6.1f calls  avg %7.1f us  %s" % (t, c, avg, nm))
