#!/usr/bin/env python
"""Round 6 debug: which autograd node breaks HIP-graph capture of a backward pass?  python tools/debug_capture2.py CASE"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import golden_util as gu
from mvp_pytorch_amd import dp, engine, hip, modeling, train
from mvp_pytorch_amd.synthetic import synthetic_batch

case = sys.argv[1]
dev = torch.device("cuda:0")
torch.manual_seed(0)


def capture(fn, warm=2):
    side = torch.cuda.Stream(dev) if os.environ.get("WARM_SIDE") else None
    if side is not None:
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(warm):
                fn()
    else:
        for _ in range(warm):
            fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    print("capturing", case, flush=True)
    with torch.cuda.graph(g, stream=side, capture_error_mode=os.environ.get("CAP_MODE", "thread_local")):
        fn()
    print("capture ended", flush=True)
    g.replay()
    torch.cuda.synchronize()
    print("replayed OK:", case, flush=True)


if case == "torch":
    lin = torch.nn.Linear(64, 64).to(dev)
    x = torch.randn(32, 64, device=dev)

    def fn():
        lin.zero_grad(set_to_none=False)
        lin(x).sum().backward()
    lin(x).sum().backward()
    capture(fn)
elif case.startswith("enc"):
    cfg = dict(gu.TINY_CFG, num_hidden_layers=2, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
    engine.DEFER_WGRAD = "nodefer" not in case
    enc = modeling.modeling_vlbert.CaptionBertEncoder(modeling.make_config(cfg)).to(dev).train()
    B, L = 8, 32
    x = (torch.randn(B * L, 128, device=dev) * 0.5).to(torch.bfloat16)
    dy = (torch.randn(B * L, 128, device=dev) * 0.1).to(torch.bfloat16)
    starts = (torch.arange(B, dtype=torch.int32) * L).to(dev)
    lens = torch.full((B,), L, dtype=torch.int32, device=dev)
    if "arena" in case:
        sync = dp.GradSync(enc)

    def fn():
        if "arena" in case:
            sync.zero_grad()
        else:
            enc.zero_grad(set_to_none=True)
        xin = x.clone().requires_grad_(True)
        y = enc.forward_rows(xin, starts, lens, B, L)
        if "fwdonly" not in case:
            y.backward(dy)
    capture(fn)
elif case == "tap":
    src = (torch.randn(300, 128, device=dev)).to(torch.bfloat16).requires_grad_(True)
    idx = torch.randint(0, 300, (50,), device=dev, dtype=torch.int32)

    def fn():
        src.grad = None
        out = engine.MultiTapFn.apply(src, None, idx)[0]
        out.float().sum().backward()
    capture(fn)
elif case.startswith("model"):
    cfg = dict(gu.TINY_CFG, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0, max_phrases=3, parallel_stacks=False)
    engine.DEFER_WGRAD = "nodefer" not in case
    dims = dict(B=16, T=12, P=3, G=6, R=5)
    batch = synthetic_batch(dims, cfg, 33, device=dev)
    batch.pop("phrase_index"); batch.pop("image_index")
    model = modeling.BiBertImgForPreTraining(modeling.make_config(cfg)).to(dev).train()
    sync = dp.GradSync(model) if "arena" in case else None

    def fn():
        if sync is not None:
            sync.zero_grad()
        else:
            model.zero_grad(set_to_none=True)
        out = model(**train.model_inputs(batch, dims["G"]))
        k = int(case.split("loss")[1][0]) if "loss" in case else 0
        out[k].backward()
    capture(fn)
