#!/usr/bin/env python
"""profiles/r02_dominant_traffic.json from the rocprofv3 --pmc passes of tools/prof_dominant.py and
tools/calib_fetch.py:  traffic_json.py OUT.json CALIB_FETCH_DIR PACKED_FETCH_DIR PACKED_WRITE_DIR FULL_FETCH_DIR FULL_WRITE_DIR"""
import csv
import glob
import json
import sys
from collections import defaultdict


def per_kernel(d, counter):
    acc = defaultdict(list)
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            if row["Counter_Name"] != counter:
                continue
            acc[row["Kernel_Name"]].append((int(row.get("Dispatch_Id", 0)), float(row["Counter_Value"])))
    return acc


def pick(acc, frag):
    vals = []
    for k, v in acc.items():
        if frag in k:
            vals += v
    return [x for _, x in sorted(vals)]


out, calib_d, pf, pw, ff, fw = sys.argv[1:7]
cal = pick(per_kernel(calib_d, "FETCH_SIZE"), "stream_read_kernel")
GiB = float(1 << 30)
# FETCH_SIZE is reported in KiB
f_lds = sum(cal[0::2]) / max(1, len(cal[0::2])) * 1024.0 / GiB
f_glb = sum(cal[1::2]) / max(1, len(cal[1::2])) * 1024.0 / GiB
res = {"_how": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, --kernel-trace only) of tools/prof_dominant.py; "
               "mean per launch over the step's launch mix.  FETCH_SIZE (KiB) is divided by the factor measured in the same "
               "session on a known 1-GiB stream through the same load path (tools/calib_fetch.py, mvptr_diag_stream_read); "
               "WRITE_SIZE is exact for 16-B stores and f32 atomics (MI355X_MICROARCH.md).",
       "fetch_size_calibration": {"lds_dma_16B_per_lane": round(f_lds, 4), "global_load_dwordx4": round(f_glb, 4),
                                  "bytes_read_per_launch": int(GiB), "launch_values_KiB": cal}}
for key, fd, wd in (("row_packed_batch", pf, pw), ("all_slots_valid", ff, fw)):
    fa, wa = per_kernel(fd, "FETCH_SIZE"), per_kernel(wd, "WRITE_SIZE")
    ent = {}
    for name, frag in (("gemm_tn_q_kernel", "gemm_tn_q_kernel"), ("gemm_tn_sk_kernel", "gemm_tn_sk_kernel"),
                       ("gemm_nt_kernel<EPI_BIAS_GELU>", "gemm_nt_kernelILi1E"), ("gemm_nt_kernel<EPI_BIAS_GELU>", "gemm_nt_kernel<1,"),
                       ("gemm_nt_kernel (all epilogues)", "gemm_nt_kernel"),
                       ("gemm_nt8_kernel<EPI_BIAS_GELU>", "gemm_nt8_kernelILi1E"), ("gemm_nt8_kernel<EPI_BIAS_GELU>", "gemm_nt8_kernel<1,"),
                       ("gemm_nt8_kernel (all epilogues)", "gemm_nt8_kernel")):
        f, w = pick(fa, frag), pick(wa, frag)
        if not f or not w:
            continue
        fr = sum(f) / len(f) * 1024.0
        wr = sum(w) / len(w) * 1024.0
        if frag == "gemm_tn_q_kernel":     # the reduce kernel over the per-split slabs belongs to the same launch
            f2, w2 = pick(fa, "tn_reduce_kernel"), pick(wa, "tn_reduce_kernel")
            if f2 and w2:
                fr += sum(f2) / len(f) * 1024.0
                wr += sum(w2) / len(w) * 1024.0
        ent[name] = {"fetch_size_raw_bytes": fr, "fetch_bytes_calibrated": fr / f_lds if f_lds > 0 else None,
                     "write_size_bytes": wr, "bytes_per_launch": round(fr / f_lds + wr) if f_lds > 0 else None,
                     "launches": len(f)}
    if key == "row_packed_batch":
        ent["rows_per_launch_group"] = [10917, 11143, 37748]   # bench.py's timed batch (seed 1234), printed by prof_dominant.py
    res[key] = ent
json.dump(res, open(out, "w"), indent=1)
print(json.dumps(res, indent=1)[:3000])
