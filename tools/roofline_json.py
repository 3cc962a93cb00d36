#!/usr/bin/env python
"""profiles/rNN_roofline.json: the dominant kernel's roofline numbers recomputed from the rocprofv3
kernel summaries of tools/prof_dominant.py (launch mix of one step) and the PMC traffic file:
    roofline_json.py OUT.json MIX_PACKED_STATS.csv MIX_FULL_STATS.csv TRAFFIC.json [BENCH_PACKED_STATS.csv BENCH_FIXED_STATS.csv]
With the two optional kernel summaries of bench.py's timed region the IN-STEP figures (same kernel between the
step's other kernels, operands from HBM) are reported beside the replay figures."""
import csv
import json
import sys

H, I = 768, 3072
PEAK = 2500.0
out, packed_csv, full_csv, traffic = sys.argv[1:5]
bench_csv = {"row_packed_batch": sys.argv[5] if len(sys.argv) > 5 else None, "all_slots_valid": sys.argv[6] if len(sys.argv) > 6 else None}
tr = json.load(open(traffic))


def avg_us(path, frag):
    """mean duration and call count over EVERY kernel whose name contains frag (template variants of one kernel —
    gemm_tn_q_kernel<4, false> / <4, true> — are one row each in the summary)"""
    tot, calls = 0.0, 0
    for r in csv.DictReader(open(path)):
        if any(f in r["Name"] for f in ((frag,) if isinstance(frag, str) else frag)):
            tot += float(r["TotalDurationNs"]) / 1e3
            calls += int(r["Calls"])
    return (tot / calls, calls) if calls else (None, 0)


def mix(Ms):
    fl, by = [], []
    for M in Ms:
        fl += [2.0 * M * 2 * H * I, 2.0 * M * 4 * H * H]
        by += [2.0 * M * (H + I) * 2 + 2 * H * I * 4, 2.0 * M * (H + H + 3 * H + H) + 4 * H * H * 4]
    return sum(fl) / len(fl), sum(by) / len(by)


res = {"_what": "gemm_tn_q_kernel<4, slab?> (grouped weight gradients, 36 launches per step; atomic write-out for the joint stack, per-split slabs + tn_reduce_kernel for the text / visual stacks), mean over the step's launch mix; "
                "duration = rocprofv3 --kernel-trace --stats AverageNs of tools/prof_dominant.py, gemm_tn_q_kernel + the tn_reduce_kernel "
                "that sums its per-split slabs (one launch = both); FLOPs = 2MNK summed over the "
                "problems of a launch; algorithmic bytes = operands read once + f32 outputs accumulated once; peak = 2500 TFLOP/s "
                "dense bf16 MFMA (MI355X_MICROARCH.md)",
       "peak_tflops": PEAK}
rows_packed = tr.get("row_packed_batch", {}).get("rows_per_launch_group", [10917, 11143, 37748])
for key, path, Ms in (("row_packed_batch", packed_csv, rows_packed), ("all_slots_valid", full_csv, [19200, 17920, 64000])):
    us, calls = avg_us(path, "gemm_tn_q_kernel")
    us_red, calls_red = avg_us(path, "tn_reduce_kernel")   # second kernel of the same launch (slab write-out)
    if us and us_red:
        us += us_red * calls_red / calls
    flop, alg = mix(Ms)
    t = tr.get(key, {}).get("gemm_tn_q_kernel", {})
    res[key] = {"rows_per_launch_group": Ms, "avg_launch_us": us, "of_which_reduce_kernel_us": (us_red * calls_red / calls) if us_red else 0.0,
                "launches_profiled": calls, "flop_per_launch": flop,
                "achieved_tflops": flop / us / 1e6 if us else None, "frac_of_peak": flop / us / 1e6 / PEAK if us else None,
                "algorithmic_bytes_per_launch": alg, "hbm_bytes_per_launch_pmc": t.get("bytes_per_launch"),
                "traffic_over_algorithmic": (t.get("bytes_per_launch") / alg) if t.get("bytes_per_launch") else None}
    if bench_csv[key]:
        us_in, calls_in = avg_us(bench_csv[key], "gemm_tn_q_kernel")
        red_in, red_calls = avg_us(bench_csv[key], "tn_reduce_kernel")
        if us_in and red_in:
            us_in += red_in * red_calls / calls_in      # the reduce kernel of the slab launches belongs to its launch
        if us_in:
            res[key]["in_step"] = {"avg_launch_us": us_in, "launches_profiled": calls_in, "achieved_tflops": flop / us_in / 1e6,
                                   "frac_of_peak": flop / us_in / 1e6 / PEAK,
                                   "source": "rocprofv3 --kernel-trace --stats of bench.py --steps 10 --warmup 3 --no-extras (timed region + warm-up)"}
# the gemm_nt_kernel family (every forward / data-gradient GEMM of the encoder layers: 8 shapes x 3 row counts in the replay)
def nt_mix(Ms):
    fl, by = [], []
    for M in Ms:
        for N, K, outs in ((3 * H, H, 1), (H, H, 2), (I, H, 1.5), (H, I, 2), (I, H, 1.5), (H, I, 2), (H, H, 1), (H, 3 * H, 2)):
            fl.append(2.0 * M * N * K)
            by.append(2.0 * M * K + 2.0 * N * K + 2.0 * M * N * outs)
    return sum(fl) / len(fl), sum(by) / len(by)


res["gemm_nt_family"] = {"_what": "gemm_nt_kernel<EPI, 64, 2, 2, 4, 8 | 6> (256 x 256 tiles; 192 x 256 where launch()'s tile-height rule picks them): the eight forward / data-gradient GEMMs of an encoder layer with their fused "
                                  "epilogues, mean over the 24 launches of the replay (8 shapes x 3 row counts); in_step: every gemm_nt_kernel of the "
                                  "traced training steps with those tile configurations (the layers' 144 launches per step + the region-embedding GEMM)"}
for key, path, Ms in (("row_packed_batch", packed_csv, rows_packed), ("all_slots_valid", full_csv, [19200, 17920, 64000])):
    us, calls = avg_us(path, "gemm_nt_kernel")
    flop, alg = nt_mix(Ms)
    t = tr.get(key, {}).get("gemm_nt_kernel (all epilogues)", {})
    ent = {"avg_launch_us": us, "launches_profiled": calls, "flop_per_launch": flop, "achieved_tflops": flop / us / 1e6 if us else None,
           "frac_of_peak": flop / us / 1e6 / PEAK if us else None, "algorithmic_bytes_per_launch": alg,
           "hbm_bytes_per_launch_pmc": t.get("bytes_per_launch"),
           "traffic_over_algorithmic": (t.get("bytes_per_launch") / alg) if t.get("bytes_per_launch") else None}
    if bench_csv[key]:
        us_in, calls_in = avg_us(bench_csv[key], ("64, 2, 2, 4, 8, 0>", "64, 2, 2, 4, 6, 0>"))   # 256- and 192-row tiles of the layer GEMMs
        if us_in:
            ent["in_step"] = {"avg_launch_us": us_in, "launches_profiled": calls_in, "achieved_tflops": flop / us_in / 1e6,
                              "frac_of_peak": flop / us_in / 1e6 / PEAK}
    res["gemm_nt_family"][key] = ent
json.dump(res, open(out, "w"), indent=1)
print(json.dumps(res, indent=1))
