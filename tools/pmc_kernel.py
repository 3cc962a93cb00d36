#!/usr/bin/env python
"""Per-dispatch values of one PMC counter for the kernels whose name contains FRAG, from a rocprofv3 --pmc output directory:
    pmc_kernel.py DIR COUNTER FRAG      (FETCH_SIZE / WRITE_SIZE are KiB; FETCH_SIZE still needs the x2 of tools/calib_fetch.py)"""
import csv
import glob
import sys

d, counter, frag = sys.argv[1:4]
vals = []
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if row["Counter_Name"] == counter and frag in row["Kernel_Name"]:
            vals.append((int(row.get("Dispatch_Id", 0)), float(row["Counter_Value"])))
vals = [v for _, v in sorted(vals)]
print("%s %s in %s: %d dispatches, mean %.1f, values %s" % (counter, frag, d, len(vals), sum(vals) / max(1, len(vals)), [round(v) for v in vals[:12]]))
