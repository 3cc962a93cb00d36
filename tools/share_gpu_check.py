#!/usr/bin/env python
"""Two INDEPENDENT processes (no torch.distributed) sharing one GPU: is the two-stream slowdown seen
in tools/dp_gloo_check.py a property of GPU sharing between processes rather than of the DP path?"""
import os
import sys
import time

import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def worker(rank, streams, q):
    import bench
    from mvp_pytorch_amd import modeling, train
    from mvp_pytorch_amd.synthetic import synthetic_batch
    dev = torch.device("cuda:0")
    cfg = dict(bench.BASE_CFG, parallel_stacks=streams)
    model = modeling.BiBertImgForPreTraining(modeling.make_config(cfg)).to(dev).train()
    opt, sched = train.build_optimizer(model, t_total=100)
    batch = synthetic_batch(dict(B=32, T=70, P=5, G=20, R=50), cfg, 7 + rank, device=dev)
    for _ in range(2):
        train.pretrain_step(model, batch, opt, sched, max_tag_length=20)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(4):
        train.pretrain_step(model, batch, opt, sched, max_tag_length=20)
    torch.cuda.synchronize()
    q.put((rank, (time.perf_counter() - t0) / 4 * 1e3))


if __name__ == "__main__":
    ctx = mp.get_context("spawn")
    for nproc in (1, 2):
        for streams in (False, True):
            q = ctx.Queue()
            ps = [ctx.Process(target=worker, args=(r, streams, q)) for r in range(nproc)]
            for p in ps:
                p.start()
            res = sorted(q.get(timeout=600) for _ in ps)
            for p in ps:
                p.join()
            print("%d process(es) on one GPU, two_streams=%s: %s ms/step" % (nproc, streams, ["%.0f" % r[1] for r in res]), flush=True)
