#!/usr/bin/env python
"""Where does a 2-rank gloo step spend its time with the text/visual stacks on two streams?"""
import os
import sys
import time

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def worker(rank, world, port, streams, hooks):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import bench
    from mvp_pytorch_amd import dp, modeling, train
    from mvp_pytorch_amd.synthetic import synthetic_batch
    dev = torch.device("cuda:0")
    torch.manual_seed(1)
    cfg = dict(bench.BASE_CFG, parallel_stacks="always" if streams else False)
    model = modeling.BiBertImgForPreTraining(modeling.make_config(cfg)).to(dev).train()
    opt, sched = train.build_optimizer(model, t_total=100)
    sync = dp.GradSync(model, overlap=False) if hooks else None
    batch = synthetic_batch(dict(B=32, T=70, P=5, G=20, R=50), cfg, 7 + rank, device=dev)
    kw = train.model_inputs(batch, 20)

    def tick():
        torch.cuda.synchronize()
        return time.perf_counter()

    for it in range(4):
        t0 = tick()
        loss = model(**kw)[0]
        t1 = tick()
        loss.backward()
        t2 = tick()
        if sync is not None:
            sync()
        t3 = tick()
        opt.step()
        sched.step()
        if sync is not None:
            sync.zero_grad()
        else:
            opt.zero_grad(set_to_none=True)
        t4 = tick()
        if rank == 0 and it >= 2:
            print("two_streams=%s gradsync_hooks=%s: fwd %.0f ms, bwd %.0f ms, exchange %.0f ms, optimizer %.0f ms"
                  % (streams, hooks, (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3, (t4 - t3) * 1e3), flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    port = 29800
    for streams, hooks in ((False, True), (True, True), (True, False)):
        port += 1
        mp.spawn(worker, args=(2, port, streams, hooks), nprocs=2, join=True)
