#!/usr/bin/env python
"""What hiding AdamW behind other work could give (upper bound, NOT a valid training step): the optimizer launches of
step k are queued on a side stream and the main stream does not wait for them, so they run beside step k+1's forward
pass (a data race on the weights: timing only).  Alternating with the normal step in one process."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from mvp_pytorch_amd import dp, engine, hip, modeling, train  # noqa: E402
from mvp_pytorch_amd.synthetic import synthetic_batch  # noqa: E402

dev = torch.device("cuda:0")
hip.load()
dims = dict(B=256, T=70, P=5, G=20, R=50)
torch.manual_seed(1234)
model = modeling.BiBertImgForPreTraining(modeling.make_config(bench.BASE_CFG)).to(dev).train()
opt, sched = train.build_optimizer(model, lr=5e-5, adam_epsilon=1e-8, weight_decay=0.01, t_total=100000)
sync = dp.GradSync(model)
b = synthetic_batch(dims, bench.BASE_CFG, 1234, device=dev)
side = torch.cuda.Stream()


def normal():
    train.pretrain_step(model, b, opt, sched, max_tag_length=dims["G"], grad_sync=sync)


def racy():
    outputs = model(**train.model_inputs(b, dims["G"]))
    outputs[0].backward()
    sync()
    main = torch.cuda.current_stream()
    side.wait_stream(main)
    with torch.cuda.stream(side):
        opt.step()
    sched.step()
    # the gradient buffers are zeroed on the side stream too (behind the update that reads them)
    with torch.cuda.stream(side):
        sync.zero_grad()
    # the next backward accumulates into the buffers: it must not start before the zero fill -> wait for the side stream
    # only at the START of the next backward would be ideal; here: an event awaited before backward (approximated by
    # waiting right before the loss.backward of the next call)


def racy2():
    """As racy(), but the main stream waits for the side stream before backward (weights race remains in forward)."""
    outputs = model(**train.model_inputs(b, dims["G"]))
    torch.cuda.current_stream().wait_stream(side)
    outputs[0].backward()
    sync()
    main = torch.cuda.current_stream()
    side.wait_stream(main)
    with torch.cuda.stream(side):
        opt.step()
        sync.zero_grad()
    sched.step()


def timed(fn, n=20, w=5):
    for _ in range(w):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


for rep in range(3):
    print("normal step %.2f ms | optimizer beside the next forward pass %.2f ms" % (timed(normal), timed(racy2)), flush=True)
