#!/usr/bin/env python
"""profiles/rNN_region_features.json from three rocprofv3 passes of tools/prof_features.py:
    features_json.py OUT.json STATS.csv FETCH_DIR WRITE_DIR CALIB_FACTOR"""
import csv
import glob
import json
import sys
from collections import defaultdict

out, stats, fetch_d, write_d, calib = sys.argv[1:6]
calib = float(calib)
rows, D, Dp, H = 256 * 50, 2054, 2056, 768


def counter(d, name, frag):
    vals = []
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == name and frag in r["Kernel_Name"]:
                vals.append(float(r["Counter_Value"]))
    return sum(vals) / len(vals) if vals else None


res = {"_what": "region-feature path of one configs[1] step (256 pairs x 50 regions x 2054 f32), each launch behind a 768-MB write; durations = "
                "rocprofv3 --kernel-trace --stats AverageNs; hbm bytes = --pmc FETCH_SIZE (KiB, divided by the calibration factor "
                "of tools/calib_fetch.py) + --pmc WRITE_SIZE (KiB), separate passes"}
for frag, key, alg_r, alg_w, flops in (("cast_rows_kernel", "cast_f32_to_bf16_kpad", rows * D * 4, rows * Dp * 2, 0.0),
                                       ("gemm_nt_kernel<0", "img_embedding_gemm_2056x768", rows * Dp * 2 + H * Dp * 2, rows * H * 2, 2.0 * rows * Dp * H)):
    us = None
    for r in csv.DictReader(open(stats)):
        if frag in r["Name"]:
            us = float(r["AverageNs"]) / 1e3
    f, w = counter(fetch_d, "FETCH_SIZE", frag), counter(write_d, "WRITE_SIZE", frag)
    e = {"avg_us": us, "algorithmic_bytes_read": alg_r, "algorithmic_bytes_written": alg_w,
         "algorithmic_TB_per_s": (alg_r + alg_w) / us / 1e6 if us else None,
         "pmc_fetch_bytes": f * 1024.0 / calib if f else None, "pmc_write_bytes": w * 1024.0 if w else None}
    if e["pmc_fetch_bytes"] and e["pmc_write_bytes"] and us:
        e["pmc_hbm_TB_per_s"] = (e["pmc_fetch_bytes"] + e["pmc_write_bytes"]) / us / 1e6
    if flops and us:
        e["tflops"] = flops / us / 1e6
    res[key] = e
json.dump(res, open(out, "w"), indent=1)
print(json.dumps(res, indent=1))
