#!/usr/bin/env python
"""Soak run: many training steps with a FRESH variable-length batch every step (new row counts, new
pack indices, new hard negatives each time).  Checks: finite losses, bounded memory, loss trend."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from mvp_pytorch_amd import dp, modeling, train  # noqa: E402
from mvp_pytorch_amd.synthetic import synthetic_batch  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 120
B = int(sys.argv[2]) if len(sys.argv) > 2 else 128
dev = torch.device("cuda:0")
dims = dict(B=B, T=70, P=5, G=20, R=50)
torch.manual_seed(0)
model = modeling.BiBertImgForPreTraining(modeling.make_config(bench.BASE_CFG)).to(dev).train()
opt, sched = train.build_optimizer(model, lr=1e-4, adam_epsilon=1e-8, weight_decay=0.01, warmup_steps=10, t_total=steps)
sync = dp.GradSync(model)      # gradient arena + fused global-norm clip (max_grad_norm below), as a training job runs it
# a small pool of distinct batches generated up front (the generator is a Python loop), cycled + re-seeded lengths
pool = [synthetic_batch(dims, bench.BASE_CFG, 1000 + i, device=dev) for i in range(12)]
hist, mem = [], []
t0 = time.time()
for s in range(steps):
    losses = train.pretrain_step(model, pool[s % len(pool)], opt, sched, max_tag_length=dims["G"], return_losses=True,
                                 grad_sync=sync, max_grad_norm=10.0)
    if s % 10 == 0 or s == steps - 1:
        vals = [float(x) for x in losses]
        hist.append((s, vals))
        mem.append(torch.cuda.max_memory_allocated() / 2 ** 30)
        assert all(v == v and abs(v) < 1e4 for v in vals), (s, vals)
        print("step %4d  total %.4f  mcp %.4f  clip %.4f  mlm %.4f  itm %.4f  wra %.4f   peak mem %.2f GiB" % ((s,) + tuple(vals) + (mem[-1],)), flush=True)
torch.cuda.synchronize()
print("done: %d steps, %.1f s, first total %.3f -> last total %.3f, peak memory %.2f GiB (after 10 steps %.2f)"
      % (steps, time.time() - t0, hist[0][1][0], hist[-1][1][0], mem[-1], mem[1]))
assert hist[-1][1][0] < hist[0][1][0], "loss did not go down"
assert mem[-1] < mem[1] * 1.15 + 0.5, "memory keeps growing"
