#!/usr/bin/env python
"""profiles/rNN_roofline.json (round 5 kernels): roofline numbers of the two GEMM kernels that lead a step, recomputed from the
rocprofv3 kernel summaries of tools/prof_dominant.py (replay of one step's launches) and the PMC traffic file:
    roofline_json.py OUT.json MIX_PACKED_STATS.csv MIX_FULL_STATS.csv TRAFFIC.json [BENCH_PACKED_STATS.csv BENCH_FIXED_STATS.csv]
With the two optional kernel summaries of bench.py's timed region the IN-STEP figures (same kernels between the step's other
kernels) are reported beside the replay figures."""
import csv
import json
import sys

H, I, LAYERS = 768, 3072, 6
PEAK = 2500.0
out, packed_csv, full_csv, traffic = sys.argv[1:5]
bench_csv = {"row_packed_batch": sys.argv[5] if len(sys.argv) > 5 else None, "all_slots_valid": sys.argv[6] if len(sys.argv) > 6 else None}
tr = json.load(open(traffic))


def avg_us(path, frag):
    """mean duration and call count over EVERY kernel whose name contains frag (template variants are one row each)"""
    tot, calls = 0.0, 0
    for r in csv.DictReader(open(path)):
        if any(f in r["Name"] for f in ((frag,) if isinstance(frag, str) else frag)):
            tot += float(r["TotalDurationNs"]) / 1e3
            calls += int(r["Calls"])
    return (tot / calls, calls) if calls else (None, 0)


def tn_mix(Ms):
    """one stack launch per row count: six layers x (FFN2, FFN1, attention output, Q/K/V) weight gradients"""
    fl = [2.0 * M * LAYERS * (2 * H * I + 4 * H * H) for M in Ms]
    by = [LAYERS * (2.0 * M * (2 * (H + I) + 6 * H) + (2 * H * I + 4 * H * H) * 4) for M in Ms]
    return sum(fl) / len(fl), sum(by) / len(by)


def nt_mix(Ms):
    fl, by = [], []
    for M in Ms:
        for N, K, outs in ((3 * H, H, 1), (H, H, 2), (I, H, 1.5), (H, I, 2), (I, H, 1.5), (H, I, 2), (H, H, 1), (H, 3 * H, 2)):
            fl.append(2.0 * M * N * K)
            by.append(2.0 * M * K + 2.0 * N * K + 2.0 * M * N * outs)
    return sum(fl) / len(fl), sum(by) / len(by)


rows_packed = tr.get("row_packed_batch", {}).get("rows_per_launch_group", [10917, 11143, 37748])
res = {"peak_tflops": PEAK,
       "gemm_nt8_family": {"_what": "gemm_nt8_kernel<EPI, MT> (ping-pong loop, 256 x 256 x 64 tiles, shorter tiles by the CU-rounds rule): the eight forward / "
                                    "data-gradient GEMMs of an encoder layer with their fused epilogues, mean over the 24 launches of the replay (8 shapes x 3 row "
                                    "counts; tools/prof_dominant.py under rocprofv3 --kernel-trace --stats); in_step: every gemm_nt8_kernel of the traced training "
                                    "steps (the layers' 144 launches per step); FLOPs = 2MNK; algorithmic bytes = operands once + every [M, N] matrix the "
                                    "epilogue moves once (8-bit stash = half); peak = 2500 TFLOP/s dense bf16 MFMA (MI355X_MICROARCH.md)"},
       "gemm_tn_stack": {"_what": "gemm_tn_sk_kernel<4>: every weight gradient of an encoder stack in one balanced launch (24 problems = 6 layers x 4; 3 launches "
                                  "per step, one per stack), mean over the step's three launches; FLOPs = 2MNK summed over the problems; algorithmic bytes = "
                                  "every operand once + the f32 results accumulated once"}}
for key, path, Ms in (("row_packed_batch", packed_csv, rows_packed), ("all_slots_valid", full_csv, [19200, 17920, 64000])):
    for name, frag, mixf, tkey in (("gemm_nt8_family", "gemm_nt8_kernel", nt_mix, "gemm_nt8_kernel (all epilogues)"),
                                   ("gemm_tn_stack", "gemm_tn_sk_kernel", tn_mix, "gemm_tn_sk_kernel")):
        us, calls = avg_us(path, frag)
        flop, alg = mixf(Ms)
        t = tr.get(key, {}).get(tkey, {})
        ent = {"rows_per_launch_group": Ms, "avg_launch_us": us, "launches_profiled": calls, "flop_per_launch": flop,
               "achieved_tflops": flop / us / 1e6 if us else None, "frac_of_peak": flop / us / 1e6 / PEAK if us else None,
               "algorithmic_bytes_per_launch": alg, "hbm_bytes_per_launch_pmc": t.get("bytes_per_launch"),
               "traffic_over_algorithmic": (t.get("bytes_per_launch") / alg) if t.get("bytes_per_launch") else None}
        if bench_csv[key]:
            us_in, calls_in = avg_us(bench_csv[key], frag)
            if us_in:
                ent["in_step"] = {"avg_launch_us": us_in, "launches_profiled": calls_in, "achieved_tflops": flop / us_in / 1e6,
                                  "frac_of_peak": flop / us_in / 1e6 / PEAK,
                                  "source": "rocprofv3 --kernel-trace --stats of bench.py --steps 10 --warmup 3 --no-extras (timed region + warm-up)"}
        res[name][key] = ent
json.dump(res, open(out, "w"), indent=1)
print(json.dumps(res, indent=1))
