#!/usr/bin/env python
"""Two ranks sharing one GPU over gloo (RCCL refuses that): step time of the data-parallel path with
the execution knobs on/off.  Only a harness check (gloo moves gradients through host memory)."""
import os
import sys
import time

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def worker(rank, world, port, unpad, streams, overlap):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import bench
    from mvp_pytorch_amd import dp, modeling, train
    from mvp_pytorch_amd.synthetic import synthetic_batch
    dev = torch.device("cuda:0")
    torch.manual_seed(1)
    cfg = dict(bench.BASE_CFG, unpad="train" if unpad else False, parallel_stacks="always" if streams else False)
    model = modeling.BiBertImgForPreTraining(modeling.make_config(cfg)).to(dev).train()
    opt, sched = train.build_optimizer(model, t_total=100)
    sync = dp.GradSync(model, overlap=overlap)
    dims = dict(B=32, T=70, P=5, G=20, R=50)
    batch = synthetic_batch(dims, cfg, 7 + rank, device=dev)
    for _ in range(2):
        train.pretrain_step(model, batch, opt, sched, max_tag_length=20, grad_sync=sync)
    torch.cuda.synchronize()
    dist.barrier()
    t0 = time.perf_counter()
    for _ in range(3):
        train.pretrain_step(model, batch, opt, sched, max_tag_length=20, grad_sync=sync)
    torch.cuda.synchronize()
    dist.barrier()
    if rank == 0:
        print("unpad=%s two_streams=%s overlap=%s: %.0f ms/step" % (unpad, streams, overlap, (time.perf_counter() - t0) / 3 * 1e3), flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    port = 29700
    for unpad, streams, overlap in ((False, False, True), (True, False, True), (True, True, True), (True, True, False)):
        port += 1
        mp.spawn(worker, args=(2, port, unpad, streams, overlap), nprocs=2, join=True)
