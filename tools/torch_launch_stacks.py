#!/usr/bin/env python
"""aten operators of one packed training step that launch device kernels, grouped by Python stack (torch.profiler
key_averages(group_by_stack_n)): where the remaining torch-side launches of the step come from."""
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
rows = []
for e in prof.key_averages(group_by_stack_n=12):
    if not e.key.startswith("aten::") or e.self_device_time_total <= 0:
        continue
    frames = [f for f in (e.stack or []) if "mvp_pytorch_amd" in f or "bench.py" in f]
    site = frames[0].split("mvp_pytorch_amd/")[-1] if frames else "(no package frame) " + " | ".join((e.stack or [])[:2])[:90]
    rows.append((e.count, e.key, site, e.self_device_time_total))
rows.sort(key=lambda r: (r[2], -r[0]))
tot = sum(r[0] for r in rows)
print("aten ops with device time in one step: %d calls" % tot)
for c, k, s, t in rows:
    print("  %3d  %-28s %7.1f us  %s" % (c, k, t, s[:120]))
