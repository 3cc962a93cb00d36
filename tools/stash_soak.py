#!/usr/bin/env python
"""the 8-bit gelu'(u) stash (round 4, |error| <= 0.0025, deterministic rounding) against the bf16 stash
of rounds 1-3 over a few thousand optimisation steps: the same BERT-base two-stage model, initial weights, batch stream,
dropout seeds and optimiser in both runs — only config.gelu_stash differs.  Prints both loss curves (means over windows of
`--window` steps), their differences, and the run-to-run difference of two bf16 runs for scale (atomically accumulated
weight gradients make two identical runs differ in the last bits, which training then amplifies).
Run on the GPU box:  python tools/stash_soak.py [--steps 3000] [--batch 32]"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from mvp_pytorch_amd import dp, engine, modeling, train  # noqa: E402
from mvp_pytorch_amd.synthetic import synthetic_batch  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=3000)
ap.add_argument("--batch", type=int, default=32)
ap.add_argument("--window", type=int, default=100)
ap.add_argument("--pool", type=int, default=48)
ap.add_argument("--runs", type=int, default=3, help="independent runs per format (their dropout / draw seeds differ: the spread between runs of ONE "
                                                   "format is the scale against which the difference between the formats is read)")
args = ap.parse_args()
dev = torch.device("cuda:0")
dims = dict(B=args.batch, T=70, P=5, G=20, R=50)
pool = [synthetic_batch(dims, bench.BASE_CFG, 5000 + i, device=dev) for i in range(args.pool)]


def run(fmt, seed_off=0):
    torch.manual_seed(0)
    engine._seed_counter[0] = 0x5DEECE66D
    model = modeling.BiBertImgForPreTraining(modeling.make_config(dict(bench.BASE_CFG, gelu_stash=fmt))).to(dev).train()
    opt, sched = train.build_optimizer(model, lr=5e-5, adam_epsilon=1e-8, weight_decay=0.01, warmup_steps=100, t_total=args.steps)
    sync = dp.GradSync(model)
    tot = torch.zeros(args.steps, device=dev)
    t0 = time.time()
    for s in range(args.steps):
        torch.manual_seed(10_000 + s + 1_000_003 * seed_off)   # same hard-negative permutation / WRA draws / dropout seeds in both formats of a run
        losses = train.pretrain_step(model, pool[s % len(pool)], opt, sched, max_tag_length=dims["G"], return_losses=True,
                                     grad_sync=sync, max_grad_norm=10.0)
        tot[s] = losses[0]
    torch.cuda.synchronize()
    sync.close()
    curve = tot.view(-1, args.window).mean(1).cpu()
    print("%-5s run %d: %d steps in %.1f s; window means: %s" % (fmt, seed_off, args.steps, time.time() - t0, " ".join("%.4f" % v for v in curve.tolist())), flush=True)
    del model, opt, sync
    torch.cuda.empty_cache()
    return curve


assert args.steps % args.window == 0
curves = {"bf16": [], "u8": []}
for r in range(args.runs):
    for fmt in ("bf16", "u8"):
        curves[fmt].append(run(fmt, r))
k = max(1, len(curves["u8"][0]) // 5)
for fmt in ("bf16", "u8"):
    fin = torch.stack([c[-k:].mean() for c in curves[fmt]])
    mid = torch.stack([c[len(c) // 2 - 1:len(c) // 2 + 1].mean() for c in curves[fmt]])
    print("%-5s mean total loss over the last fifth of the run, per run: %s -> mean %.4f, std %.4f; at mid-run: mean %.4f, std %.4f"
          % (fmt, " ".join("%.4f" % v for v in fin.tolist()), fin.mean(), fin.std(unbiased=True) if len(fin) > 1 else 0.0, mid.mean(),
             mid.std(unbiased=True) if len(mid) > 1 else 0.0))
pair = torch.stack([(a[-k:].mean() - b[-k:].mean()) for a, b in zip(curves["u8"], curves["bf16"])])
print("u8 - bf16 per run (same seeds within a run), last fifth: %s -> mean %.4f" % (" ".join("%+.4f" % v for v in pair.tolist()), pair.mean()))
