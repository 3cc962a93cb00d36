#!/usr/bin/env python
"""Run-to-run reproducibility of one pre-training step's gradients (same seeds, same process): one / two
streams, with / without WRA."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import golden_util as gu  # noqa: E402
from mvp_pytorch_amd import modeling  # noqa: E402
from mvp_pytorch_amd.synthetic import synthetic_batch  # noqa: E402

dev = torch.device("cuda:0")
cfg = dict(gu.TINY_CFG, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
dims = dict(B=12, T=20, P=4, G=8, R=9)
b = {k: v.to(dev) for k, v in synthetic_batch(dims, cfg, 31).items()}
import itertools


def run(fast, streams, wra):
    torch.manual_seed(17)
    model = modeling.BiBertImgForPreTraining(modeling.make_config(cfg)).to(dev).train()
    model.wra_on_device = True
    model.bert.parallel_stacks = streams
    for enc in (model.bert.txt_encoder, model.bert.vis_encoder, model.bert.mul_encoder):
        enc.unpad = True
    torch.manual_seed(123)
    o = model(input_ids_a=b["input_ids_a"], token_type_ids_a=b["segment_ids_a"], attention_mask_a=b["input_mask_a"],
              masked_lm_labels_a=b["lm_label_ids_a"], input_ids_b=b["input_ids_b"], img_feats=b["img_feats"],
              token_type_ids_b=b["segment_ids_b"], attention_mask_b=b["input_mask_b"],
              masked_lm_labels_b=b["lm_label_ids_b"], max_tag_length=dims["G"], phrase_index=b["phrase_index"] if wra else None,
              img_index=b["image_index"] if wra else None)
    o[0].backward()
    torch.cuda.synchronize()
    return {n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None}


for fast, streams, wra in itertools.product((False,), (True, False), (True, False)):
    g1, g2 = run(fast, streams, wra), run(fast, streams, wra)
    bad = [n for n in g1 if not torch.equal(g1[n], g2[n])]
    groups = sorted(set(n.split(".")[1] if n.startswith("bert.") else n.split(".")[0] for n in bad))
    print("two_streams=%s wra=%s: %d of %d gradients differ run to run; in: %s" % (streams, wra, len(bad), len(g1), groups))
