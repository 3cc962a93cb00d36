#!/usr/bin/env python
"""Retrieval re-ranking throughput (BASELINE configs[3] shape family: README retrieval lengths
max_seq_length 50 + 5 phrases, 30 tag slots, 50 regions => La=55, Lb=80, Lj=105): per-pair
forward_mod='fine' as the reference evaluates (both uni-modal encoders recomputed for every pair,
run_retrieval.py:755-790) vs the cached two-stage engine (encode each caption / image once, then
mul_encoder only)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from mvp_pytorch_amd import modeling  # noqa: E402
from mvp_pytorch_amd.synthetic import synthetic_batch  # noqa: E402

dev = torch.device("cuda:0")
n_img, caps_per_img, topk = 200, 5, 64
dims = dict(B=n_img * caps_per_img, T=50, P=5, G=30, R=50)
cfg = dict(bench.BASE_CFG, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0, loss_type="ce", num_labels=2)
torch.manual_seed(0)
model = modeling.BiImageBertForRetrieval(modeling.make_config(cfg)).to(dev).eval()
b = synthetic_batch(dims, cfg, 7, device=dev)
n_txt = dims["B"]
img_of = torch.arange(n_txt, device=dev) // caps_per_img      # caption i describes image i // 5
img_rows = torch.arange(0, n_txt, caps_per_img, device=dev)    # one copy of every image


def sync_time(fn):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = fn()
    torch.cuda.synchronize()
    return out, time.perf_counter() - t0


def encode_all(packed=True):
    text = {k: [] for k in ("seq", "mask", "glob")}
    for s0 in range(0, n_txt, 500):
        sl = slice(s0, s0 + 500)
        t = model.encode_text(input_ids_a=b["input_ids_a"][sl], token_type_ids_a=b["segment_ids_a"][sl], attention_mask_a=b["input_mask_a"][sl], packed=packed)
        for k in text:
            text[k].append(t[k])
    text = {k: torch.cat(v) for k, v in text.items()}
    r = img_rows
    image = model.encode_image(input_ids_b=b["input_ids_b"][r], img_feats=b["img_feats"][r], token_type_ids_b=b["segment_ids_b"][r],
                               attention_mask_b=b["input_mask_b"][r], max_tag_length=dims["G"], packed=packed)
    return text, image


(text, image), t_enc = sync_time(encode_all)
sim = model.coarse_scores(text, image)                       # [n_txt, n_img]
cand = sim.topk(topk, dim=1).indices                         # text -> top-k images to re-rank
ti = torch.arange(n_txt, device=dev).repeat_interleave(topk)
ii = cand.reshape(-1)
n_pairs = ti.numel()
model.rerank(text, image, ti[:4096], ii[:4096])              # warm-up
scores, t_rr = sync_time(lambda: model.rerank(text, image, ti, ii, chunk=4096))
text_pad, image_pad = encode_all(packed=False)
scores_pad, t_rr_pad = sync_time(lambda: model.rerank(text_pad, image_pad, ti, ii, chunk=4096, packed=False))

# reference-style: materialise every pair and run the whole Bi model on it
model.forward_mod = "fine"


def fine_all(limit):
    out = []
    for s0 in range(0, limit, 1024):
        t, i = ti[s0:s0 + 1024], img_rows[ii[s0:s0 + 1024]]
        with torch.no_grad():
            out.append(model(input_ids_a=b["input_ids_a"][t], token_type_ids_a=b["segment_ids_a"][t], attention_mask_a=b["input_mask_a"][t],
                             input_ids_b=b["input_ids_b"][i], token_type_ids_b=b["segment_ids_b"][i], attention_mask_b=b["input_mask_b"][i],
                             img_feats=b["img_feats"][i], max_tag_length=dims["G"]))
    return torch.cat(out)


sub = min(n_pairs, 16384)
fine_all(1024)
ref, t_fine = sync_time(lambda: fine_all(sub))
p_ref = torch.softmax(ref.float(), -1)[:, 1]
p_pk = torch.softmax(scores[:sub].float(), -1)[:, 1]
print("pairs %d (top-%d of %d images for %d captions); encode once %.3f s; cached rerank: row-packed %.3f s = %.0f pairs/s, "
      "padded %.3f s = %.0f pairs/s; per-pair 'fine' forward on %d pairs %.3f s = %.0f pairs/s; end-to-end speed-up %.2fx; "
      "padded cache == 'fine' bit for bit: %s; row-packed vs 'fine': max |delta p(match)| = %.2e"
      % (n_pairs, topk, n_img, n_txt, t_enc, t_rr, n_pairs / t_rr, t_rr_pad, n_pairs / t_rr_pad, sub, t_fine, sub / t_fine,
         (n_pairs / (t_rr + t_enc)) / (sub / t_fine), bool(torch.equal(scores_pad[:sub], ref)), float((p_pk - p_ref).abs().max())))
