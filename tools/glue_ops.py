#!/usr/bin/env python
"""Device time of the torch operators (not the HIP library's kernels) in one pre-training step, by
operator and input shapes: what the "torch glue" of a step consists of."""
import os
import sys

import torch
from torch.profiler import ProfilerActivity, profile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from mvp_pytorch_amd import hip, modeling, train  # noqa: E402
from mvp_pytorch_amd.synthetic import synthetic_batch  # noqa: E402

dev = torch.device("cuda:0")
hip.load()
dims = dict(B=256, T=70, P=5, G=20, R=50)
torch.manual_seed(1234)
model = modeling.BiBertImgForPreTraining(modeling.make_config(bench.BASE_CFG)).to(dev).train()
opt, sched = train.build_optimizer(model, lr=5e-5, adam_epsilon=1e-8, weight_decay=0.01, t_total=100000)
b = synthetic_batch(dims, bench.BASE_CFG, 1234, device=dev)
for _ in range(3):
    train.pretrain_step(model, b, opt, sched, max_tag_length=dims["G"])
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    train.pretrain_step(model, b, opt, sched, max_tag_length=dims["G"])
    torch.cuda.synchronize()
ev = prof.key_averages(group_by_input_shape=True)
dt = lambda e: getattr(e, "self_device_time_total", getattr(e, "self_cuda_time_total", 0))  # noqa: E731
rows = [(dt(e), e.count, e.key, str(e.input_shapes)[:110]) for e in ev if dt(e) > 0]
tot = sum(r[0] for r in rows)
print("device time of all torch-visible ops in one step: %.2f ms" % (tot / 1e3))
for t, n, k, sh in sorted(rows, reverse=True)[:45]:
    print("  %8.1f us  x%-4d %-42s %s" % (t, n, k[:42], sh))
