#!/usr/bin/env python
"""GPU idle share of a bench run from a rocprofv3 kernel trace: union of the kernel intervals against the
span of the last N steps' kernels:  gpu_idle.py KERNEL_TRACE.csv"""
import csv
import sys

rows = []
for r in csv.DictReader(open(sys.argv[1])):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
# steps are delimited by the AdamW launches (two parameter groups per step)
adam = [i for i, r in enumerate(rows) if "adamw_multi_kernel" in r[2]]
steps = adam[1::2]
if len(steps) < 6:
    raise SystemExit("not enough steps in the trace")
lo, hi = steps[-6], steps[-1]          # five whole steps
seg = rows[lo + 1:hi + 1]
t0, t1 = seg[0][0], max(e for _, e, _ in seg)
busy, cur_s, cur_e = 0, None, None
for s, e, _ in seg:
    if cur_e is None or s > cur_e:
        if cur_e is not None:
            busy += cur_e - cur_s
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
span = t1 - t0
named = sorted(((seg[i + 1][0] - max(e for _, e, _ in seg[max(0, i - 8):i + 1]), seg[i][2][:60], seg[i + 1][2][:60])
                for i in range(len(seg) - 1)), reverse=True)
gaps = [g for g, _, _ in named]
print("5 steps: span %.2f ms (%.2f ms/step), GPU busy %.2f ms = %.1f %%, idle %.2f ms/step, kernels/step %d, sum of kernel time %.2f ms/step"
      % (span / 1e6, span / 5e6, busy / 1e6, 100.0 * busy / span, (span - busy) / 5e6, len(seg) // 5,
         sum(e - s for s, e, _ in seg) / 5e6))
print("largest gaps (us):", [round(g / 1e3, 1) for g in gaps[:12]])
for g, a, b in named[:14]:
    print("  %7.1f us  after %-60s before %s" % (g / 1e3, a, b))
# context of the largest gap: the launches either side of it with their offsets from the gap's start
big = max(range(len(seg) - 1), key=lambda i: seg[i + 1][0] - max(e for _, e, _ in seg[max(0, i - 8):i + 1]))
ref = seg[big][1]
for s, e, name in seg[max(0, big - 10):big + 14]:
    print("  start %+9.1f us  dur %7.1f us  %s" % ((s - ref) / 1e3, (e - s) / 1e3, name[:110]))

# where in the step the idle time sits: 40 equal windows over the mean step, idle us per step in each and the
# kernel that runs longest in the window
W = 40
step_bounds = [rows[steps[k]][0] for k in range(-6, 0)]
idle = [0.0] * W
names = [dict() for _ in range(W)]
for k in range(5):
    a, b = step_bounds[k], step_bounds[k + 1]
    ks = [r for r in rows if a <= r[0] < b]
    cur = a
    for s_, e_, n_ in ks:
        w = min(W - 1, int((s_ - a) * W / (b - a)))
        if s_ > cur:
            idle[w] += (s_ - cur) / 5e3
        cur = max(cur, e_)
        names[w][n_[:70]] = names[w].get(n_[:70], 0) + (e_ - s_)
print("idle time along the step (window = 1/%d of the step; us of idle per step in the window; busiest kernel):" % W)
for w in range(W):
    top = max(names[w].items(), key=lambda kv: kv[1])[0] if names[w] else "-"
    print("  %2d  idle %7.1f us   %s" % (w, idle[w], top))

# steady state (the same five steps): launches and kernel time per step, this library's kernels against everything else
ours = lambda n: ("anonymous namespace" in n or "_GLOBAL__N_" in n) and "at::native" not in n   # noqa: E731
agg = {}
for s_, e_, n_ in seg:
    k = n_[:96]
    a = agg.setdefault(k, [0, 0])
    a[0] += 1
    a[1] += e_ - s_
mine = [(v[1] / 5e6, v[0] / 5.0, k) for k, v in agg.items() if ours(k)]
other = [(v[1] / 5e6, v[0] / 5.0, k) for k, v in agg.items() if not ours(k)]
print("steady state per step: %d launches, %.2f ms of kernel time; this library %d launches / %.2f ms; torch, rocBLAS and "
      "runtime copies %d launches / %.3f ms" % (len(seg) // 5, sum(e - s for s, e, _ in seg) / 5e6, round(sum(c for _, c, _ in mine)),
                                                sum(t for t, _, _ in mine), round(sum(c for _, c, _ in other)), sum(t for t, _, _ in other)))
for t, c, k in sorted(mine, reverse=True)[:24]:
    print("   %7.3f ms %6.1f x  %s" % (t, c, k))
print("  not this library:")
for t, c, k in sorted(other, reverse=True)[:40]:
    print("   %7.3f ms %6.1f x  %s" % (t, c, k))
