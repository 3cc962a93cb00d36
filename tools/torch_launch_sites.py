#!/usr/bin/env python
"""Every device launch of one packed training step that does NOT come from libmvptr_hip.so, by aten operator and by the innermost
frame of this package that asked for it (torch.profiler with stacks): the worklist for the launch count of the step's torch glue."""
import collections
import os
import sys

import torch
from torch.profiler import ProfilerActivity, profile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from mvp_pytorch_amd import dp, hip, modeling, train  # noqa: E402
from mvp_pytorch_amd.synthetic import synthetic_batch  # noqa: E402

dev = torch.device("cuda:0")
hip.load()
dims = dict(B=256, T=70, P=5, G=20, R=50)
torch.manual_seed(1234)
model = modeling.BiBertImgForPreTraining(modeling.make_config(bench.BASE_CFG)).to(dev).train()
opt, sched = train.build_optimizer(model, lr=5e-5, adam_epsilon=1e-8, weight_decay=0.01, t_total=100000)
sync = dp.GradSync(model)
b = synthetic_batch(dims, bench.BASE_CFG, 1234, device=dev)
for _ in range(3):
    train.pretrain_step(model, b, opt, sched, max_tag_length=dims["G"], grad_sync=sync, max_grad_norm=10.0)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    train.pretrain_step(model, b, opt, sched, max_tag_length=dims["G"], grad_sync=sync, max_grad_norm=10.0)
    torch.cuda.synchronize()
sites = collections.Counter()
total = 0
for e in prof.events():
    ks = [k for k in (getattr(e, "kernels", None) or []) if "anonymous namespace" not in k.name and "_GLOBAL__N" not in k.name]
    if not ks or not e.name.startswith("aten::"):
        continue
    # count a kernel once: at the innermost aten op that owns it (children are listed too, so skip ops with aten children that own the kernels)
    if any(c.name.startswith("aten::") and (getattr(c, "kernels", None) or []) for c in (e.cpu_children or [])):
        continue
    st = [f for f in (e.stack or []) if "mvp_pytorch_amd" in f or "bench.py" in f]
    par, p = "", e.cpu_parent
    while p is not None:
        if "Backward" in p.name or p.name.endswith("Fn") or "Optimizer" in p.name:
            par = p.name
            break
        p = p.cpu_parent
    site = (st[0].split("mvp_pytorch_amd/")[-1] if st else "(autograd engine)") + ("  [" + par + "]" if par else "")
    sites[(site, e.name)] += len(ks)
    total += len(ks)
print("torch-side device launches of one step: %d" % total)
by_site = collections.Counter()
for (site, op), c in sites.items():
    by_site[site] += c
for site, c in by_site.most_common(70):
    ops = ", ".join("%s x%d" % (op.replace("aten::", ""), n) for (s2, op), n in sorted(sites.items(), key=lambda kv: -kv[1]) if s2 == site)
    print("  %3d  %s   { %s }" % (c, site[:110], ops[:160]))
