#!/usr/bin/env python
"""The two fused AdamW launches of the pre-training step (mvptr_adamw_multi: parameters without bf16 working copies;
mvptr_adamw_mirror_multi: encoder / head weights, update + bf16 copies): elements, HBM bytes and time per launch."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from mvp_pytorch_amd import dp, engine, hip, modeling, train  # noqa: E402
from mvp_pytorch_amd.synthetic import synthetic_batch  # noqa: E402

dev = torch.device("cuda:0")
dims = dict(B=64, T=70, P=5, G=20, R=50)
model = modeling.BiBertImgForPreTraining(modeling.make_config(bench.BASE_CFG)).to(dev).train()
opt, sched = train.build_optimizer(model, lr=5e-5, adam_epsilon=1e-8, weight_decay=0.01, t_total=100000)
sync = dp.GradSync(model)
b = synthetic_batch(dims, bench.BASE_CFG, 1, device=dev)
for _ in range(3):
    train.pretrain_step(model, b, opt, sched, max_tag_length=dims["G"], grad_sync=sync)
torch.cuda.synchronize()
log = []


def timed(name, fn, elems):
    def w(*a, **k):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        r = fn(*a, **k)
        e1.record()
        log.append((name, elems(*a), e0, e1))
        return r
    return w


n_plain = lambda tab, ct, co, n, *a: n      # noqa: E731  (chunks)
hip_multi, hip_mirror = hip.adamw_multi, hip.adamw_mirror_multi
hip.adamw_multi = timed("adamw_multi", hip_multi, lambda *a: a[3])
hip.adamw_mirror_multi = timed("adamw_mirror_multi", hip_mirror, lambda *a: a[3] if len(a) > 3 else 0)
for _ in range(5):
    train.pretrain_step(model, b, opt, sched, max_tag_length=dims["G"], grad_sync=sync)
torch.cuda.synchronize()
plain = sum(p.numel() for p in model.parameters() if engine.mirror_of(p) is None and p.requires_grad)
mirrored = sum(p.numel() for p in model.parameters() if engine.mirror_of(p) is not None)
print("parameters without working copies: %.2f M (28 B each per step), with: %.2f M (32-34 B each)" % (plain / 1e6, mirrored / 1e6))
per = {}
for name, units, e0, e1 in log[-4:]:
    print("  %-20s %7d chunks/tiles  %7.1f us" % (name, units, e0.elapsed_time(e1) * 1e3))
    per[name] = per.get(name, 0.0) + e0.elapsed_time(e1) * 1e3
print("adamw_multi %.1f us/step = %.2f TB/s; adamw_mirror_multi %.1f us/step = %.2f TB/s (32 B per element)"
      % (per["adamw_multi"], plain * 28 / per["adamw_multi"] / 1e6, per["adamw_mirror_multi"], mirrored * 32 / per["adamw_mirror_multi"] / 1e6))
