#!/usr/bin/env python
"""Which call sites launch the fill (zero) kernels of a packed training step?  torch.profiler with Python stacks: every
aten::zero_ / aten::fill_ / aten::zeros of one step grouped by the innermost frame inside this package (autograd-engine
fills — gradients of unused outputs, AccumulateGrad — show up under their backward node)."""
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
want = sys.argv[1].split(",") if len(sys.argv) > 1 else ["aten::zero_", "aten::fill_", "aten::zeros", "aten::zeros_like", "aten::new_zeros", "aten::full"]
sites = collections.Counter()
kern = collections.Counter()
for e in prof.events():
    if e.device_type == torch.autograd.DeviceType.CUDA or str(e.device_type).endswith("CUDA"):
        kern[e.name[:90]] += 1
        continue
    if e.name in want:
        st = [f for f in (e.stack or []) if "mvp_pytorch_amd" in f or "bench.py" in f]
        par, p = "", e.cpu_parent
        while p is not None:
            if "Backward" in p.name or "Fn" in p.name or "Optimizer" in p.name:
                par = p.name
                break
            p = p.cpu_parent
        sites[(e.name, (st[0].split("mvp_pytorch_amd/")[-1] if st else "(no python frame)") + ("  [" + par + "]" if par else ""))] += 1
print("fill-like aten ops of one step by call site:")
for (n, s), c in sites.most_common(60):
    print("  %3d  %-18s %s" % (c, n, s))
print("device kernels of the step, top by count:")
for n, c in kern.most_common(25):
    print("  %4d  %s" % (c, n))
