#!/usr/bin/env python
"""Where do the small fill / zero kernels of a training step come from?  Counts call sites of
torch.zeros / zeros_like / Tensor.zero_ / new_zeros / torch.full during one step."""
import collections
import os
import sys
import traceback

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from mvp_pytorch_amd import modeling, train  # noqa: E402
from mvp_pytorch_amd.synthetic import synthetic_batch  # noqa: E402

dev = torch.device("cuda:0")
dims = dict(B=256, T=70, P=5, G=20, R=50)
model = modeling.BiBertImgForPreTraining(modeling.make_config(bench.BASE_CFG)).to(dev)
model.train()
opt, sched = train.build_optimizer(model, lr=5e-5, adam_epsilon=1e-8, weight_decay=0.01, t_total=1000)
batch = synthetic_batch(dims, bench.BASE_CFG, 1, device=dev)
for _ in range(2):
    train.pretrain_step(model, batch, opt, sched, max_tag_length=dims["G"])
counts = collections.Counter()


def wrap(mod, name):
    orig = getattr(mod, name)

    def f(*a, **k):
        st = traceback.extract_stack(limit=4)[:-1]
        key = name + " <- " + " <- ".join("%s:%d" % (os.path.basename(fr.filename), fr.lineno) for fr in reversed(st))
        counts[key] += 1
        return orig(*a, **k)
    setattr(mod, name, f)


for n in ("zeros", "zeros_like", "full", "ones", "eye", "arange"):
    wrap(torch, n)
for n in ("zero_", "new_zeros", "fill_"):
    wrap(torch.Tensor, n)
train.pretrain_step(model, batch, opt, sched, max_tag_length=dims["G"])
torch.cuda.synchronize()
for k, v in counts.most_common(40):
    print("%4d  %s" % (v, k))
print("total", sum(counts.values()))
