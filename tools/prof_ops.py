#!/usr/bin/env python
"""Which torch ops (outside the HIP library) take GPU time in a training step?  torch.profiler, one step."""
import os
import sys

import torch
from torch.profiler import ProfilerActivity, profile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from mvp_pytorch_amd import modeling, train  # noqa: E402
from mvp_pytorch_amd.synthetic import synthetic_batch  # noqa: E402

dev = torch.device("cuda:0")
dims = dict(B=256, T=70, P=5, G=20, R=50)
model = modeling.BiBertImgForPreTraining(modeling.make_config(bench.BASE_CFG)).to(dev).train()
opt, sched = train.build_optimizer(model, lr=5e-5, adam_epsilon=1e-8, weight_decay=0.01, t_total=1000)
batch = synthetic_batch(dims, bench.BASE_CFG, 1234, device=dev)
for _ in range(3):
    train.pretrain_step(model, batch, opt, sched, max_tag_length=dims["G"])
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=False) as prof:
    train.pretrain_step(model, batch, opt, sched, max_tag_length=dims["G"])
    torch.cuda.synchronize()
rows = []
for e in prof.key_averages():
    t = getattr(e, "self_device_time_total", None)
    if t is None:
        t = getattr(e, "self_cuda_time_total", 0)
    if t > 0:
        rows.append((t, e.count, e.key))
rows.sort(reverse=True)
tot = sum(r[0] for r in rows)
print("GPU time by op (self), total %.2f ms" % (tot / 1e3))
for t, n, k in rows[:45]:
    print("%8.1f us  x%4d  %s" % (t, n, k[:100]))
