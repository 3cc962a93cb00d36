#!/usr/bin/env python
"""Round 6 debug: which part of a training step breaks HIP-graph capture?  One stage per process (a crash in hipStreamEndCapture
takes the process down):  python tools/debug_capture.py STAGE [--one-stream] [--no-wra] [--base]
stages: salt | fwd | fwdbwd | clip | full"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import golden_util as gu
from mvp_pytorch_amd import dp, hip, modeling, train
from mvp_pytorch_amd.optimization import AdamW, WarmupLinearSchedule
from mvp_pytorch_amd.synthetic import synthetic_batch

stage = sys.argv[1]
dev = torch.device("cuda:0")
base = "--base" in sys.argv
cfg = dict(gu.BASE_CFG if base else gu.TINY_CFG, hidden_dropout_prob=0.1, attention_probs_dropout_prob=0.1, max_phrases=3)
if "--one-stream" in sys.argv:
    cfg["parallel_stacks"] = False
dims = dict(B=16, T=12, P=3, G=6, R=5)
batch = synthetic_batch(dims, cfg, 33, device=dev)
if "--no-wra" in sys.argv:
    batch.pop("phrase_index"); batch.pop("image_index")
torch.manual_seed(0)
model = modeling.BiBertImgForPreTraining(modeling.make_config(cfg)).to(dev).train()
opt = AdamW([{"params": list(model.parameters()), "weight_decay": 0.01}], lr=1e-3, eps=1e-8)
sched = WarmupLinearSchedule(opt, warmup_steps=10, t_total=100)
sync = dp.GradSync(model)
for _ in range(2):
    train.pretrain_step(model, batch, opt, sched, max_tag_length=dims["G"], grad_sync=sync, max_grad_norm=1.0)
torch.cuda.synchronize()
static = {k: (v.clone() if isinstance(v, torch.Tensor) else v) for k, v in batch.items()}
salt = hip.dropout_salt(dev)
g = torch.cuda.CUDAGraph()
plan = []
print("capturing stage", stage, sys.argv[2:], flush=True)
with torch.cuda.graph(g, capture_error_mode="thread_local"):
    salt.add_(1)
    if stage != "salt":
        if stage == "fwd":
            with torch.no_grad():
                out = model(**train.model_inputs(static, dims["G"]))
        else:
            out = model(**train.model_inputs(static, dims["G"]))
            out[0].backward()
            if stage in ("clip", "full"):
                sync(want_norm=True)
                coef = train.clip_coefficient(model, sync, 1.0)
            if stage == "full":
                opt._graph_plan = plan
                opt.step(grad_scale=coef)
                opt._graph_plan = None
            sync.zero_grad()
print("capture ended", flush=True)
g.replay()
torch.cuda.synchronize()
print("replayed OK: stage", stage, sys.argv[2:], flush=True)
