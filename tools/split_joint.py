#!/usr/bin/env python
"""Joint stack (6 layers) forward + backward over the matched and the hard pairs: ONE 2n-sequence pass (what the
model does) against TWO n-sequence passes on two streams (the reference's two passes, run concurrently)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from mvp_pytorch_amd import engine, modeling  # noqa: E402

dev = torch.device("cuda:0")
torch.manual_seed(0)
cfg = modeling.make_config(bench.BASE_CFG)
model = modeling.BiBertImgForPreTraining(cfg).to(dev).train()
enc = model.bert.mul_encoder
n, Lj, H = 256, 125, 768
fixed = len(sys.argv) > 1 and sys.argv[1] == "fixed"


def make(nseq, seed):
    g = torch.Generator().manual_seed(seed)
    lens = torch.full((nseq,), Lj) if fixed else torch.randint(21, 126, (nseq,), generator=g)
    mask = (torch.arange(Lj)[None, :] < lens[:, None]).float()
    add = ((1.0 - mask) * -10000.0).to(dev)
    x = (torch.randn(nseq, Lj, H, generator=g) * 0.5).to(torch.bfloat16).to(dev)
    return x, add


xa, ma = make(n, 1)
xb, mb = make(n, 2)
x2, m2 = torch.cat([xa, xb]), torch.cat([ma, mb])


def merged():
    x = x2.clone().requires_grad_(True)
    y = enc(x, m2)[0]
    y.float().square().mean().backward()


def split():
    main = torch.cuda.current_stream()
    side = engine.side_stream(dev)
    a = xa.clone().requires_grad_(True)
    b = xb.clone().requires_grad_(True)
    side.wait_stream(main)
    with torch.cuda.stream(side):
        yb = enc(b, mb)[0]
        lb = yb.float().square().mean()
    ya = enc(a, ma)[0]
    la = ya.float().square().mean()
    main.wait_stream(side)
    (la + lb).backward()


def timeit(fn, reps=8):
    for _ in range(3):
        fn()
        enc.zero_grad(set_to_none=True)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
        enc.zero_grad(set_to_none=True)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


for name, fn in (("one 2n pass", merged), ("two n passes, two streams", split), ("one 2n pass", merged), ("two n passes, two streams", split)):
    print("%s (%s): %.2f ms forward + backward" % (name, "all slots valid" if fixed else "variable lengths", timeit(fn)))
