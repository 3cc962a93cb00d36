#!/usr/bin/env python
"""Host time spent inside train.pretrain_step (kernel enqueue + Python) against the wall time per
step: how close the step is to being launch-bound."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from mvp_pytorch_amd import hip, modeling, train  # noqa: E402
from mvp_pytorch_amd.synthetic import synthetic_batch  # noqa: E402

dev = torch.device("cuda:0")
hip.load()
fixed = len(sys.argv) > 1 and sys.argv[1] == "fixed"
dims = dict(B=256, T=70, P=5, G=20, R=50)
torch.manual_seed(1234)
model = modeling.BiBertImgForPreTraining(modeling.make_config(bench.BASE_CFG)).to(dev).train()
opt, sched = train.build_optimizer(model, lr=5e-5, adam_epsilon=1e-8, weight_decay=0.01, t_total=100000)
from mvp_pytorch_amd import dp  # noqa: E402
sync = dp.GradSync(model)          # the gradient arena, as bench.py runs the step
b = synthetic_batch(dims, bench.BASE_CFG, 1234, fixed_length=fixed, device=dev)
for _ in range(3):
    train.pretrain_step(model, b, opt, sched, max_tag_length=dims["G"], grad_sync=sync)
torch.cuda.synchronize()
host, n = 0.0, 10
t0 = time.perf_counter()
for _ in range(n):
    h0 = time.perf_counter()
    train.pretrain_step(model, b, opt, sched, max_tag_length=dims["G"], grad_sync=sync)
    host += time.perf_counter() - h0
torch.cuda.synchronize()
wall = time.perf_counter() - t0
print("%s: wall %.2f ms/step, host time inside pretrain_step %.2f ms/step (includes waiting at the step's host syncs)"
      % ("fixed" if fixed else "packed", wall / n * 1e3, host / n * 1e3))
# the same with an EMPTY queue in front of every step (no back-pressure from a full launch queue can hide in the figure)
solo = []
for _ in range(8):
    torch.cuda.synchronize()
    h0 = time.perf_counter()
    train.pretrain_step(model, b, opt, sched, max_tag_length=dims["G"], grad_sync=sync)
    solo.append((time.perf_counter() - h0) * 1e3)
torch.cuda.synchronize()
print("host time of one step queued behind an empty queue (ms): %s" % [round(v, 2) for v in solo])
# where the host waits: the row-count read-backs (engine.AsyncCounts.get)
from mvp_pytorch_amd import engine  # noqa: E402
waits = []
_get = engine.AsyncCounts.get


def timed_get(self):
    t = time.perf_counter()
    r = _get(self)
    waits.append(time.perf_counter() - t)
    return r


engine.AsyncCounts.get = timed_get
for _ in range(5):
    train.pretrain_step(model, b, opt, sched, max_tag_length=dims["G"], grad_sync=sync)
torch.cuda.synchronize()
per = len(waits) // 5
print("count read-backs per step: %d; host wait in each (us, mean of 5 steps): %s"
      % (per, [round(sum(waits[i::per]) / 5 * 1e6) for i in range(per)]))
engine.AsyncCounts.get = _get
import cProfile  # noqa: E402
import pstats  # noqa: E402
pr = cProfile.Profile()
pr.enable()
for _ in range(5):
    train.pretrain_step(model, b, opt, sched, max_tag_length=dims["G"], grad_sync=sync)
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr, stream=sys.stdout)
print("cProfile, 5 steps, by own time:")
st.sort_stats("tottime").print_stats(45)
print("cProfile, 5 steps, by cumulative time:")
st.sort_stats("cumulative").print_stats(60)
from torch.profiler import profile, ProfilerActivity  # noqa: E402
with profile(activities=[ProfilerActivity.CPU]) as prof:
    train.pretrain_step(model, b, opt, sched, max_tag_length=dims["G"], grad_sync=sync)
    torch.cuda.synchronize()
ev = prof.key_averages()
tot = sum(e.self_cpu_time_total for e in ev) / 1e3
print("torch profiler: self CPU time of all torch ops in one step %.2f ms; top ops:" % tot)
for e in sorted(ev, key=lambda e: -e.self_cpu_time_total)[:14]:
    print("   %-50s calls %5d  self cpu %.2f ms" % (e.key[:50], e.count, e.self_cpu_time_total / 1e3))

# who asks for zero fills / small copies: Python entry points wrapped for one step, grouped by calling frame
import collections  # noqa: E402
import traceback  # noqa: E402
sites = collections.Counter()


def _wrap(owner, name):
    orig = getattr(owner, name)

    def w(*a, **k):
        st = traceback.extract_stack(limit=6)[:-1]
        fr = next((f for f in reversed(st) if "mvp_pytorch_amd" in f.filename or f.filename.endswith("bench.py")), st[-1])
        sites[(name, "%s:%d %s" % (fr.filename.split("mvp_pytorch_amd/")[-1], fr.lineno, fr.name))] += 1
        return orig(*a, **k)
    setattr(owner, name, w)
    return orig


saved = [(o, n, _wrap(o, n)) for o, n in ((torch, "zeros"), (torch, "zeros_like"), (torch, "full"), (torch, "ones"), (torch, "cat"),
                                          (torch, "arange"), (torch, "stack"), (torch.Tensor, "zero_"), (torch.Tensor, "fill_"),
                                          (torch.Tensor, "copy_"), (torch.Tensor, "to"), (torch.Tensor, "float"),
                                          (torch.Tensor, "contiguous"), (torch.Tensor, "index_select"), (torch.Tensor, "sum"))]
train.pretrain_step(model, b, opt, sched, max_tag_length=dims["G"], grad_sync=sync)
torch.cuda.synchronize()
for o, n, f in saved:
    setattr(o, n, f)
print("python-level tensor constructors / small ops of one step by call site:")
for (name, frame), c in sorted(sites.items(), key=lambda kv: -kv[1])[:70]:
    print("   %4d  %-12s %s" % (c, name, frame))
