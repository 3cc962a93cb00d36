#!/usr/bin/env python
"""Diagnostic (not collected by pytest): packed vs padded execution of the configs[1]-shape step, every
parameter gradient, against the f32 CPU oracle on the same weights / inputs / hard batch.  Answers
whether a packed-vs-padded difference on one parameter is arithmetic noise (both far from the oracle
by the same amount) or a defect (one of them close, the other not).

    python tests/diag_cfg1_grads.py [B]
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)
import golden_util as gu  # noqa: E402
from oracle import mvptr_oracle as orc  # noqa: E402
from test_model_gpu import Replay  # noqa: E402


def rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return ((a - b).norm() / (b.norm() + 1e-30)).item()


def main():
    from mvp_pytorch_amd import modeling, train
    from mvp_pytorch_amd.synthetic import synthetic_batch
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    dev = torch.device("cuda:0")
    cfg = dict(gu.BASE_CFG, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0, max_phrases=5)
    dims = dict(gu.CFG2_DIMS, B=B)
    b = synthetic_batch(dims, cfg, 77)
    bd = {k: v.to(dev) for k, v in b.items()}
    perm = torch.randperm(B, generator=torch.Generator().manual_seed(5))
    torch.manual_seed(0)
    ref_model = modeling.BiBertImgForPreTraining(modeling.make_config(cfg))
    sd = {k: v.clone() for k, v in ref_model.state_dict().items()}
    # oracle with gradients (CPU f32), WRA left out on both sides (phrase / image index not passed)
    sdg = {k: (v.clone().requires_grad_(True) if v.is_floating_point() else v) for k, v in sd.items()}
    torch.set_num_threads(min(64, os.cpu_count() or 8))
    res_o, aux = orc.bi_bert_img_for_pretraining(
        sdg, cfg, b["input_ids_a"], b["segment_ids_a"], b["input_mask_a"], b["lm_label_ids_a"], b["input_ids_b"],
        b["segment_ids_b"], b["input_mask_b"], b["lm_label_ids_b"], dims["G"], b["img_feats"],
        draws=orc.Draws(randperm=[perm.numpy()]), return_aux=True)
    res_o[0].backward()
    masked = aux["sim_mat"].detach() - 2 * torch.eye(B)
    hard = (masked.max(1)[1], masked.max(0)[1])
    print("oracle losses", [float(x) for x in res_o])
    grads = {}
    for unpad in (True, False):
        model = modeling.BiBertImgForPreTraining(modeling.make_config(cfg))
        model.load_state_dict(sd)
        model.to(dev).train()
        model.wra_on_device = False
        for enc in (model.bert.txt_encoder, model.bert.vis_encoder, model.bert.mul_encoder):
            enc.unpad = unpad
        kw = train.model_inputs(bd, dims["G"])
        kw.pop("phrase_index", None)
        kw.pop("img_index", None)
        with Replay(dict(draw_randperm=[perm.numpy()]), dev), gu.InjectHard(model.bert, hard[0], hard[1]):
            o = model(**kw)
        o[0].backward()
        torch.cuda.synchronize()
        print("unpad", unpad, "losses", [float(x) for x in o])
        grads[unpad] = {n: p.grad.detach().float().cpu() for n, p in model.named_parameters() if p.grad is not None}
        del model
    rows = []
    for n in grads[True]:
        go = sdg[n].grad
        if go is None or go.norm() < 1e-9:
            continue
        rows.append((rel(grads[True][n], grads[False][n]), rel(grads[True][n], go), rel(grads[False][n], go), n))
    rows.sort(reverse=True)
    print("%-70s %10s %10s %10s" % ("parameter", "pack~pad", "pack~orc", "pad~orc"))
    for r in rows[:25]:
        print("%-70s %10.2e %10.2e %10.2e" % (r[3], r[0], r[1], r[2]))
    a = np.array([[r[0], r[1], r[2]] for r in rows])
    print("median", np.median(a, axis=0), "max", a.max(axis=0))


if __name__ == "__main__":
    main()
