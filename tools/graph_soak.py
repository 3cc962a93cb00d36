#!/usr/bin/env python
"""Round 6: does a training run made of REPLAYED steps (train.GraphedStep: one HIP graph per batch signature, AdamW tables refreshed by
the host, dropout salt bumped per replay) train like the same run with every step queued from Python?  BERT-base two-stage model,
a small pool of fixed-signature batches (every slot valid and the same number of scored rows, so that all of them replay ONE graph),
lr warm-up + decay, dropout 0.1, clip 10; window means of the total loss for eager and replayed runs with the same seeds, twice each
(two eager runs differ through atomically accumulated gradients; the replayed runs additionally draw other dropout masks / hard-negative
splits than the eager ones from step 3 on — same distributions).  Run on the GPU box:  python tools/graph_soak.py [--steps 1500]"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from mvp_pytorch_amd import dp, engine, hip, modeling, train  # noqa: E402
from mvp_pytorch_amd.synthetic import synthetic_batch  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=1500)
ap.add_argument("--batch", type=int, default=32)
ap.add_argument("--window", type=int, default=100)
ap.add_argument("--pool", type=int, default=6)
args = ap.parse_args()
dev = torch.device("cuda:0")
dims = dict(B=args.batch, T=70, P=5, G=20, R=50)
# batches of one signature: all slots valid; keep generating until `pool` of them share the scored-row counts of the first
pool, seed = [], 7000
while len(pool) < args.pool and seed < 9000:
    b = synthetic_batch(dims, bench.BASE_CFG, seed, fixed_length=True, device=dev)
    if not pool or dict(b["host_counts"]) == dict(pool[0]["host_counts"]):
        pool.append(b)
    seed += 1
print("pool of %d batches with host counts %s" % (len(pool), dict(pool[0]["host_counts"])), flush=True)


def run(graphed, rep):
    torch.manual_seed(rep)
    engine._seed_counter[0] = 0x5DEECE66D + rep
    hip.dropout_salt(dev).zero_()
    model = modeling.BiBertImgForPreTraining(modeling.make_config(bench.BASE_CFG)).to(dev).train()
    opt, sched = train.build_optimizer(model, lr=5e-5, adam_epsilon=1e-8, weight_decay=0.01, warmup_steps=100, t_total=args.steps)
    sync = dp.GradSync(model)
    step = train.GraphedStep(model, opt, sched, max_tag_length=dims["G"], max_grad_norm=10.0, grad_sync=sync, enabled=graphed)
    tot = torch.zeros(args.steps, device=dev)
    t0 = time.time()
    for s in range(args.steps):
        tot[s] = step(pool[s % len(pool)])
    torch.cuda.synchronize()
    dt = time.time() - t0
    hip.check_device_errors(dev)
    sync.close()
    curve = tot.view(-1, args.window).mean(1).cpu()
    print("%-6s run %d: %d steps in %.1f s (%d replayed, %d eager, %d captures%s); window means: %s" % (
        "graph" if graphed else "eager", rep, args.steps, dt, step.replays, step.eager_steps, step.captures,
        "" if step.last_error is None else ", capture error: " + step.last_error, " ".join("%.3f" % v for v in curve.tolist())), flush=True)
    del model, opt, sync, step
    torch.cuda.empty_cache()
    return curve


assert args.steps % args.window == 0
curves = {True: [], False: []}
for rep in range(2):
    for graphed in (False, True):
        curves[graphed].append(run(graphed, rep))
k = max(1, len(curves[True][0]) // 5)
for g in (False, True):
    fin = [float(c[-k:].mean()) for c in curves[g]]
    print("%s: mean total loss over the last fifth, per run: %s" % ("graph" if g else "eager", " ".join("%.4f" % v for v in fin)))
print("graph - eager (same seeds), last fifth: %s;  eager run 1 - eager run 0: %+.4f" % (
    " ".join("%+.4f" % float(a[-k:].mean() - b[-k:].mean()) for a, b in zip(curves[True], curves[False])),
    float(curves[False][1][-k:].mean() - curves[False][0][-k:].mean())))
hip.dropout_salt(dev).zero_()
